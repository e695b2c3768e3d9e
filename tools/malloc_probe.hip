// Profiling aid: what one process pays for hipMalloc of the window kernel's work space.  Usage: malloc_probe <GB> <pieces> [hold_ms]
// Prints the wall time of the allocation(s), of a first-touch kernel over them, and of hipFree.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <thread>
__global__ void touch(unsigned long long *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x * 512) p[i] = i; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 47; const int pieces = argc > 2 ? atoi(argv[2]) : 1; const int hold = argc > 3 ? atoi(argv[3]) : 0;
  double t0 = now(); hipFree(nullptr); hipStream_t st; hipStreamCreate(&st); double t1 = now();
  std::vector<void *> ps((size_t)pieces); const size_t each = (gb << 30) / (size_t)pieces;
  for (int i = 0; i < pieces; ++i) if (hipMalloc(&ps[(size_t)i], each) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  double t2 = now();
  for (int i = 0; i < pieces; ++i) hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, st, (unsigned long long *)ps[(size_t)i], each / 8);
  hipStreamSynchronize(st); double t3 = now();
  if (hold) std::this_thread::sleep_for(std::chrono::milliseconds(hold));
  double t4 = now();
  for (int i = 0; i < pieces; ++i) hipFree(ps[(size_t)i]);
  double t5 = now();
  printf("%zu GB in %d piece(s): init %.3f s  hipMalloc %.3f s  first touch (1/512 of the words) %.3f s  hipFree %.3f s\n", gb, pieces, t1 - t0, t2 - t1, t3 - t2, t5 - t4);
  return 0;
}

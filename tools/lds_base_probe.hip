// Tuning aid (GPU box): where in a CU's LDS a workgroup's allocation starts, as the workgroup itself can read it (HW_REG_LDS_ALLOC).
//   hipcc --offload-arch=gfx950 -O2 tools/lds_base_probe.hip -o build/lds_base_probe && build/lds_base_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void __launch_bounds__(512) probe(unsigned *out, int spin) {
  extern __shared__ unsigned char lds[];
  if (threadIdx.x == 0) {
    const unsigned a = __builtin_amdgcn_s_getreg(6 | (31 << 11));       // HW_REG_LDS_ALLOC, all 32 bits
    const unsigned h = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_REG_HW_ID
    const unsigned x = __builtin_amdgcn_s_getreg(20 | (31 << 11));      // HW_REG_XCC_ID
    out[3 * blockIdx.x] = a; out[3 * blockIdx.x + 1] = h; out[3 * blockIdx.x + 2] = x;
    lds[0] = (unsigned char)a;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(64);      // stay resident so that every CU takes its two
  }
  __syncthreads();
}
int main() {
  const int n = 512; unsigned *d; hipMalloc(&d, 12 * n); hipMemset(d, 0, 12 * n);
  for (int bytes : {81920, 10240}) {
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(probe, dim3(bytes > 20000 ? n : 4080 > n ? n : n), dim3(bytes > 20000 ? 512 : 64), bytes, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(3 * n); hipMemcpy(h.data(), d, 12 * n, hipMemcpyDeviceToHost);
    std::map<unsigned, int> hist; for (int i = 0; i < n; ++i) hist[h[3 * i]]++;
    printf("dynamic LDS %d bytes: distinct HW_REG_LDS_ALLOC values %zu\n", bytes, hist.size());
    int k = 0; for (auto &kv : hist) { if (k++ < 24) printf("  0x%08x (base field [7:0] %u, size field [20:12] %u) x %d\n", kv.first, kv.first & 0xFF, (kv.first >> 12) & 0x1FF, kv.second); }
    for (int i = 0; i < 6; ++i) printf("  wg %d: lds_alloc 0x%08x hw_id 0x%08x xcc 0x%x\n", i, h[3 * i], h[3 * i + 1], h[3 * i + 2]);
  }
  return 0;
}

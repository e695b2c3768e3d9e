// window_fat.hip -- the window kernel compiled a second time for the re-run tier: LC_FAT_LANES lanes per window
// instead of 64.  Same source (kernels.h) with LANCET_FAT defined: wave 0 runs every phase as in the one-wave kernel,
// the other waves join the streaming passes of the general build (XG_FOR, wave.h).  Only windows that overflowed the
// ordinary work space get here -- coverage pile-ups, whose cost is one build over ~10^6 k-mer occurrences.
#define LANCET_FAT 1
#include <hip/hip_runtime.h>
#include "kernels.h"

__global__ void __launch_bounds__(LC_FAT_LANES) window_kernel_fat(const lancet_params *P, const DevBatch *B, const EngineCaps *C, Work *works, DevOut *OUT) {
  window_kernel_body((LC_GLOBAL const lancet_params *)P, (LC_GLOBAL const DevBatch *)B, (LC_GLOBAL const EngineCaps *)C, (LC_GLOBAL Work *)works, (LC_GLOBAL DevOut *)OUT, (LC_WS *)&lc_shared, (int)blockIdx.x);
}

int lc_launch_window_fat(int slots, hipStream_t stream, const lancet_params *P, const DevBatch *B, const EngineCaps *C, Work *works, DevOut *OUT) {
  hipLaunchKernelGGL(window_kernel_fat, dim3(slots), dim3(LC_FAT_LANES), 0, stream, P, B, C, works, OUT);
  return (int)hipGetLastError();
}

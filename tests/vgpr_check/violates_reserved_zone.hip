// Deliberately violates the reserved-register scheme: the kernel keeps ~230 accumulators live,
// so hipcc allocates VGPRs well above v192 while the asm load below targets v[192:195].
#include <hip/hip_runtime.h>
#include "async_stage.h"
extern "C" __global__ void __launch_bounds__(64) neg_reserved_zone_kernel(const float* x, float* y, int n) {
  float acc[230];
#pragma unroll
  for (int i = 0; i < 230; ++i) acc[i] = x[threadIdx.x + 64 * i];
  TMGCN_Q_LOAD(192, 193, 194, 195, x + 4 * threadIdx.x);
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 230; ++i) acc[i] = fmaf(acc[i], acc[(i + 1) % 230], 1.0f);
  }
  TMGCN_WAIT_VM(0);
  float q[4];
  TMGCN_Q_READ(192, 193, 194, 195, q);
  float s = q[0] + q[1] + q[2] + q[3];
#pragma unroll
  for (int i = 0; i < 230; ++i) s += acc[i];
  y[threadIdx.x] = s;
}

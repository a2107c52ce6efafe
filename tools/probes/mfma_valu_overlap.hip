// Does an exact-f32 MFMA chain share its SIMD's issue/datapath with another wave's VALU work?  (round 6: the fused SpMM+GEMM
// kernel's gather and product phases ADD UP even on separate waves — DESIGN.md §4.)
// One 512-thread block per CU = two waves per SIMD.  Wave 0-3 run a dependent chain of MFMAs (f32 32x32x2 or bf16 32x32x16),
// waves 4-7 a chain of v_fma_f32 (or idle, or also MFMA).  Times: MFMA alone, VALU alone, both.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o build/mfma_valu_overlap && build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE_A, int MODE_B>   // role of waves 0-3 / 4-7: 0 idle, 1 f32 MFMA chain, 2 bf16 MFMA chain, 3 VALU fma chain
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const int mode = wave < 4 ? MODE_A : MODE_B;
  float r = 0.f;
  if (mode == 1) {
    f32x16 acc = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    r = acc[0] + acc[7];
  } else if (mode == 2) {
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(threadIdx.x * 1e-3f + k); b[k] = (__bf16)1.0f; }
    for (int i = 0; i < 2 * iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);   // 32 cycles each: same pipe time as `iters` f32 MFMAs
    r = acc[0] + acc[7];
  } else if (mode == 3) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;      // four independent chains: the VALU issues every cycle it can
    const float m = 1.0000001f, c = 1e-7f;
    for (int i = 0; i < (3 * iters) / 2; ++i) {   // ~ the f32 MFMA chain's duration
      x0 = fmaf(x0, m, c); x1 = fmaf(x1, m, c); x2 = fmaf(x2, m, c); x3 = fmaf(x3, m, c);
    }
    r = x0 + x1 + x2 + x3;
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int A, int B>
static float run(float* out, int cus, int iters) {
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {          // the first launches ramp the clock: best of six
    (void)hipEventRecord(s);
    hipLaunchKernelGGL((probe<A, B>), dim3(cus), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s, e);
    if (rep && ms < best) best = ms;
  }
  return best;
}

int main() {
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out;
  (void)hipMalloc(&out, 4096);
  const int iters = 200000;    // 200 000 f32 MFMAs x 64 cycles = 12.8 M cycles per SIMD (5.5 ms)
  printf("{\"cus\": %d, \"iters\": %d,\n", cus, iters);
  printf(" \"f32_mfma_alone_ms\": %.3f,\n", run<1, 0>(out, cus, iters));
  printf(" \"bf16_mfma_alone_ms\": %.3f,\n", run<2, 0>(out, cus, iters));
  printf(" \"valu_alone_ms\": %.3f,\n", run<0, 3>(out, cus, iters));
  printf(" \"f32_mfma_beside_valu_ms\": %.3f,\n", run<1, 3>(out, cus, iters));
  printf(" \"bf16_mfma_beside_valu_ms\": %.3f,\n", run<2, 3>(out, cus, iters));
  printf(" \"f32_mfma_beside_f32_mfma_ms\": %.3f,\n", run<1, 1>(out, cus, iters));
  printf(" \"valu_beside_valu_ms\": %.3f\n}\n", run<3, 3>(out, cus, iters));
  return 0;
}

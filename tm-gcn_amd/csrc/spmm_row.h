// Row gather shared by the plain and the GEMM-fused SpMM kernels (gfx950).
#pragma once
#include "common.h"

namespace tmgcn {

// Outputs are written once and not read again by this kernel: non-temporal stores (-1.5 % on the fused kernel,
// round 1).  The gathered X rows and the (col, val) stream stay on the default policy: non-temporal LOADS of the X
// rows are 12-15 % slower (the Infinity Cache re-use of the live gather window is real).
__device__ __forceinline__ void store_f4(float4* p, const float4& v) {
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
  __builtin_nontemporal_store(v.z, &p->z);
  __builtin_nontemporal_store(v.w, &p->w);
}
__device__ __forceinline__ void store_f1(float* p, float v) { __builtin_nontemporal_store(v, p); }

// One wave sums one CSR row:  acc = sum_p val[p] * Xs[col[p]]  over [beg, end).
// LPR lanes cover the F4 float4s of a feature row (lane fl), S = 64/LPR streams split the
// non-zeros; the row's (col,val) pairs are fetched 64 at a time with one coalesced load per
// array and handed to the streams with ds_bpermute (__shfl); U 16-B gathers in flight per lane.
// On return every lane of stream 0 (sub == 0) holds the full sum (fixed butterfly order).
template <int LPR, int U>
__device__ __forceinline__ float4 gather_row(const int32_t* __restrict__ col,
                                             const float* __restrict__ val,
                                             const float4* __restrict__ Xs, int64_t beg,
                                             int64_t end, int F4, int lane) {
  constexpr int S = kWave / LPR;
  const int sub = lane / LPR;
  const int fl = lane % LPR;
  const bool f_ok = fl < F4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t base = beg; base < end; base += kWave) {
    const int n = (int)((end - base) < kWave ? (end - base) : kWave);
    int c = 0;
    float v = 0.f;
    if (lane < n) {
      c = col[base + lane];
      v = val[base + lane];
    }
    for (int p = 0; p < n; p += S * U) {
      float4 x[U];
      float vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = p + u * S + sub;
        const int cc = __shfl(c, idx & 63);
        vv[u] = __shfl(v, idx & 63);
        x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < n && f_ok) x[u] = Xs[(int64_t)cc * F4 + fl];
        if (idx >= n) vv[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc.x = fmaf(vv[u], x[u].x, acc.x);
        acc.y = fmaf(vv[u], x[u].y, acc.y);
        acc.z = fmaf(vv[u], x[u].z, acc.z);
        acc.w = fmaf(vv[u], x[u].w, acc.w);
      }
    }
  }
#pragma unroll
  for (int o = LPR; o < kWave; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o);
    acc.y += __shfl_xor(acc.y, o);
    acc.z += __shfl_xor(acc.z, o);
    acc.w += __shfl_xor(acc.w, o);
  }
  return acc;
}

}  // namespace tmgcn

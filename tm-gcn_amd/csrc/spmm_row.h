// Row gather shared by the plain and the GEMM-fused SpMM kernels (gfx950).
#pragma once
#include "common.h"

// tuning knobs for the A/B harness (tools/ab_variants.sh)
#ifndef TMGCN_NT_COLVAL
#define TMGCN_NT_COLVAL 0   // 1: stream (col,val) with non-temporal loads
#endif
#ifndef TMGCN_NT_GATHER
#define TMGCN_NT_GATHER 0   // 1: gather the X rows with non-temporal loads (A/B: see DESIGN.md §4)
#endif
#ifndef TMGCN_NT_STORE
#define TMGCN_NT_STORE 1    // 1: non-temporal stores for the SpMM / fused outputs (A/B: -1.5 % on the fused kernel)
#endif

namespace tmgcn {

__device__ __forceinline__ void store_f4(float4* p, const float4& v) {
#if TMGCN_NT_STORE
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
  __builtin_nontemporal_store(v.z, &p->z);
  __builtin_nontemporal_store(v.w, &p->w);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void store_f1(float* p, float v) {
#if TMGCN_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// One wave sums one CSR row:  acc = sum_p val[p] * Xs[col[p]]  over [beg, end).
// LPR lanes cover the F4 float4s of a feature row (lane fl), S = 64/LPR streams split the
// non-zeros; the row's (col,val) pairs are fetched 64 at a time with one coalesced load per
// array and handed to the streams with ds_bpermute (__shfl); U 16-B gathers in flight per lane.
// On return every lane of stream 0 (sub == 0) holds the full sum (fixed butterfly order).
// PRE: the row's first 64 (col, val) pairs were fetched by the caller one row ahead (c0, v0: lane i
// holds entry beg + i, anything for i >= end - beg) — the fused kernel's software prefetch.
template <int LPR, int U, bool PRE = false>
__device__ __forceinline__ float4 gather_row(const int32_t* __restrict__ col,
                                             const float* __restrict__ val,
                                             const float4* __restrict__ Xs, int64_t beg,
                                             int64_t end, int F4, int lane, int c0 = 0, float v0 = 0.f) {
  constexpr int S = kWave / LPR;
  const int sub = lane / LPR;
  const int fl = lane % LPR;
  const bool f_ok = fl < F4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t base = beg; base < end; base += kWave) {
    const int n = (int)((end - base) < kWave ? (end - base) : kWave);
    int c = 0;
    float v = 0.f;
    if (PRE && base == beg) {
      c = c0;
      v = v0;
    } else if (lane < n) {
#if TMGCN_NT_COLVAL
      c = __builtin_nontemporal_load(col + base + lane);
      v = __builtin_nontemporal_load(val + base + lane);
#else
      c = col[base + lane];
      v = val[base + lane];
#endif
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
#if TMGCN_NT_GATHER
        if (idx < n && f_ok) {
          typedef float g_f4 __attribute__((ext_vector_type(4)));
          const g_f4 t = __builtin_nontemporal_load(reinterpret_cast<const g_f4*>(Xs + (int64_t)cc * F4 + fl));
          x[u] = make_float4(t.x, t.y, t.z, t.w);
        }
#else
        if (idx < n && f_ok) x[u] = Xs[(int64_t)cc * F4 + fl];
#endif
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

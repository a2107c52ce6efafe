// Layers 1 + 2 of the narrow 2-layer models in one forward and one backward kernel   (gfx950 / CDNA4)
//
//     Y = act1(H · W1)                  ehf:330-335 (EmbeddingGCN2, H = the cached AtXt), ehf:486 (EmbeddingKWGCN, H = AX)
//     Z = act2((Â ⋆ Y) · W2)            ehf:348-349 (the as-run default branch: the TRAINING adjacency), ehf:487
// for the reference's own widths (F0 = 2 -> 6 -> 6): H [R][2] is a constant of the model (cached at
// construction, ehf:293 / 464), so Y never has to exist.  At these sizes (R = T·N = 570 k rows, 3 non-zeros per
// row) the separate kernels are bound by the [R][6] tensors they pass to each other, not by arithmetic:
//   forward   the SpMM gathers 8-byte H rows instead of 24-byte Y rows and applies W1 and the non-linearity to
//             each gathered row on the fly (12 fmas + 6 activations per non-zero) — the per-lane fmaf chains of
//             gemm_small and spmm_gemm_small (a row's partial sums are folded over however many lanes each kernel
//             gives a row, so the results agree to the last bit or two, not always bitwise);
//   backward  dY = (Âᵀ ⋆ dZ) · W2ᵀ per row, P = H·W1 recomputed (12 fmas), dP = dY ⊙ act1'(P), and
//             dW1 = Σ_r H[r]ᵀ·dP[r] accumulated on the spot (fp64, dealt over the lanes of a row, slabs reduced by the
//             last block): no dY, no pre-activation and no dP tensor, no separate dW1 launch.
// dW2 = (Â⋆Y)ᵀ·dZ stays the narrow dW kernel (gemm.hip) on the Â⋆Y the forward stores for it (folding its 36 sums into
// the backward kernel as well was measured: 72 us instead of 37 + 13 — 48 fp64 accumulators per lane; not kept).
#include "common.h"

namespace tmgcn {

struct L12Args {
  const int64_t* rowptr;   // forward: Â; backward: Âᵀ
  const int32_t* col;
  const float* val;
  const float* H;          // [R][KI]
  const float* W1;         // [KI][F]
  const float* W2;         // [F][NT]
  const float* dZ;         // backward: [R][NT]
  const float* pre2;       // backward, optional: pre-activation of layer 2 (act2 != none)
  float* Z;                // forward: [R][NT]
  float* AX;               // forward, optional: Â⋆Y [R][F]
  float* pre2_out;         // forward, optional
  float* dW1;              // backward: [KI][F]
  float* part;             // backward: [blocks][KI·F] slabs
  int32_t* sync;
  int64_t n_rows;
  int32_t N;
  int32_t act1, act2;
};

template <int N, typename T>
__device__ __forceinline__ T pick_at(const T (&v)[N], int i) {
  T r = v[0];
#pragma unroll
  for (int q = 1; q < N; ++q) r = (i == q) ? v[q] : r;
  return r;
}

// y[f] = act1(Σ_k h[k]·W1[k][f]) — gemm_small's chain (k ascending from 0), so the same bits
template <int KI, int F>
__device__ __forceinline__ void layer1_row(const float (&h)[KI], const float (&W1)[KI][F], const ActApply& act1, float (&y)[F]) {
#pragma unroll
  for (int f = 0; f < F; ++f) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KI; ++k) s = fmaf(h[k], W1[k][f], s);
    y[f] = act1(s);
  }
}

template <int KI, int F, int NT, int G>
__global__ __launch_bounds__(256) void l12_fwd_kernel(L12Args a) {
  // uniform operands first (scalar registers), before any store of this kernel
  float W1[KI][F], W2[F][NT];
#pragma unroll
  for (int k = 0; k < KI; ++k)
#pragma unroll
    for (int f = 0; f < F; ++f) W1[k][f] = a.W1[k * F + f];
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int n = 0; n < NT; ++n) W2[f][n] = a.W2[f * NT + n];
  const ActApply act1(a.act1), act2(a.act2);
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = gid / G;
  const int gl = (int)(gid % G);
  const bool live = r < a.n_rows;
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  if (live) {
    const int64_t beg = a.rowptr[r], end = a.rowptr[r + 1];
    const int64_t xoff = (r / a.N) * (int64_t)a.N;
    // NB non-zeros of the row per trip: their (col, val) pairs are requested together and then their H rows together —
    // two dependent round trips per NB non-zeros instead of two per non-zero (these kernels are bound by that chain,
    // not by bytes).  Positions past the row's end are clamped to its last entry and carry weight 0: every load is
    // unconditional (a load under a per-lane condition becomes a branch with a full wait).  Same fmaf order per lane.
    constexpr int NB = 4;
    for (int64_t p = beg + gl; p < end; p += NB * G) {
      float v[NB];
      int c[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int64_t q = p + u * G;
        const int64_t qc = q < end ? q : end - 1;
        const float vv = a.val[qc];
        c[u] = a.col[qc];
        v[u] = q < end ? vv : 0.f;
      }
      float2 hv[NB];
      static_assert(KI == 2, "H rows are float2");
#pragma unroll
      for (int u = 0; u < NB; ++u) hv[u] = *reinterpret_cast<const float2*>(a.H + (xoff + c[u]) * KI);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (p + u * G < end) {                                   // slots past the row's end: skipped (no layer-1 work for them)
          const float h[KI] = {hv[u].x, hv[u].y};
          float y[F];
          layer1_row<KI, F>(h, W1, act1, y);
#pragma unroll
          for (int f = 0; f < F; ++f) acc[f] = fmaf(v[u], y[f], acc[f]);      // spmm_gemm_small's accumulation
        }
      }
    }
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] += __shfl_xor(acc[f], o);
  if (!live) return;
  if (a.AX) {
    for (int f = gl; f < F; f += G) a.AX[r * F + f] = pick_at<F>(acc, f);
  }
  for (int n = gl; n < NT; n += G) {
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float w = W2[f][0];
#pragma unroll
      for (int q = 1; q < NT; ++q) w = (n == q) ? W2[f][q] : w;
      s = fmaf(acc[f], w, s);
    }
    if (a.pre2_out) a.pre2_out[r * NT + n] = s;
    a.Z[r * NT + n] = act2(s);
  }
}

// Backward.  Groups of G lanes walk rows r = group, group + n_groups, …; the KI·F fp64 accumulators of dW1 are dealt
// over the lanes of a group (lane gl owns q = gl + j·G).
constexpr int kL12MaxBlocks = 1024;

template <int KI, int F, int NT, int G>
__global__ __launch_bounds__(256) void l12_bwd_kernel(L12Args a) {
  constexpr int NO = KI * F;
  constexpr int NPL = (NO + G - 1) / G;
  __shared__ double red[4][NO];
  __shared__ int is_last;
  float W1[KI][F], W2[F][NT];
#pragma unroll
  for (int k = 0; k < KI; ++k)
#pragma unroll
    for (int f = 0; f < F; ++f) W1[k][f] = a.W1[k * F + f];
#pragma unroll
  for (int f = 0; f < F; ++f)
#pragma unroll
    for (int n = 0; n < NT; ++n) W2[f][n] = a.W2[f * NT + n];
  const ActGrad dact1(a.act1), dact2(a.act2);
  const int gl = threadIdx.x & (G - 1);
  const int64_t n_groups = (int64_t)gridDim.x * 256 / G;
  double acc[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) acc[j] = 0.0;
  for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G; r < a.n_rows; r += n_groups) {
    const int64_t beg = a.rowptr[r], end = a.rowptr[r + 1];
    const int64_t xoff = (r / a.N) * (int64_t)a.N;
    float t[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) t[n] = 0.f;
    constexpr int NB = 4;                                    // non-zeros per trip, loads unconditional (see the forward kernel)
    for (int64_t p = beg + gl; p < end; p += NB * G) {
      float v[NB];
      int64_t c[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int64_t q = p + u * G;
        const int64_t qc = q < end ? q : end - 1;
        const float vv = a.val[qc];
        c[u] = xoff + a.col[qc];
        v[u] = q < end ? vv : 0.f;
      }
      float g[NB][NT];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const float2* gz = reinterpret_cast<const float2*>(a.dZ + c[u] * NT);
#pragma unroll
        for (int i = 0; i < NT / 2; ++i) {
          const float2 q = gz[i];
          g[u][2 * i] = q.x;
          g[u][2 * i + 1] = q.y;
        }
      }
      if (a.pre2) {                                          // act2 != none: dZ ⊙ act2'(pre2) of the gathered rows
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const float2* pz = reinterpret_cast<const float2*>(a.pre2 + c[u] * NT);
#pragma unroll
          for (int i = 0; i < NT / 2; ++i) {
            const float2 q = pz[i];
            g[u][2 * i] *= dact2(q.x);
            g[u][2 * i + 1] *= dact2(q.y);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u)
        if (p + u * G < end) {
#pragma unroll
          for (int n = 0; n < NT; ++n) t[n] = fmaf(v[u], g[u][n], t[n]);
        }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int n = 0; n < NT; ++n) t[n] += __shfl_xor(t[n], o);
    // dY = t·W2ᵀ (input index ascending, as the transposed-weight form of spmm_gemm_small), P = H·W1, dP = dY ⊙ act1'(P)
    const float2 hv = *reinterpret_cast<const float2*>(a.H + r * KI);
    const float h[KI] = {hv.x, hv.y};
    float dP[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) s = fmaf(t[n], W2[f][n], s);
      float pf = 0.f;
#pragma unroll
      for (int k = 0; k < KI; ++k) pf = fmaf(h[k], W1[k][f], pf);
      dP[f] = s * dact1(pf);
    }
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int q = gl + j * G;
      if (q < NO) {
        const int k = q / F, f = q - k * F;
        acc[j] = fma((double)pick_at<KI>(h, k), (double)pick_at<F>(dP, f), acc[j]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    double v = acc[j];
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) v += __shfl_xor(v, o);
    const int q = gl + j * G;
    if (lane < G && q < NO) red[wave][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < NO)
    __hip_atomic_store(reinterpret_cast<unsigned*>(a.part) + (int64_t)blockIdx.x * NO + threadIdx.x,
                       __float_as_uint((float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x])),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!last_block_ticket(a.sync, (int)gridDim.x, &is_last)) return;
  if (threadIdx.x == 0) *a.sync = 0;
  // the last block: 256 / NO threads per output, each adding its share of the slabs in order (8 loads in flight)
  constexpr int SUBS = 256 / NO;
  __shared__ double fin[SUBS][NO];
  const unsigned* P = reinterpret_cast<const unsigned*>(a.part);
  const int sub = threadIdx.x / NO, o = threadIdx.x - sub * NO;
  const int nb = (int)gridDim.x;
  if (sub < SUBS) {
    double s[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = 0.0;
    for (int c = sub; c < nb; c += 8 * SUBS) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cc = c + q * SUBS;
        const float v = __uint_as_float(__hip_atomic_load(P + (int64_t)(cc < nb ? cc : nb - 1) * NO + o, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT));
        s[q] += cc < nb ? (double)v : 0.0;
      }
    }
    fin[sub][o] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  }
  __syncthreads();
  if (threadIdx.x < NO) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < SUBS; ++q) tot += fin[q][threadIdx.x];
    a.dW1[threadIdx.x] = (float)tot;
  }
}

// lanes per row: chains of about four non-zeros per lane, walked NB at a time (kernel durations under rocprofv3, captured
// S1 / S3 steps: 3 nnz/row: G = 1 beats 2 by 18 %; 27 nnz/row: G = 4 — 41 us backward against 48 with one lane per row)
static int l12_lanes(float avg_nnz_per_row) {
  int G = 8;
  if (avg_nnz_per_row >= 0.f) {
    G = 1;
    while (G < 16 && 4.f * G <= avg_nnz_per_row) G <<= 1;
  }
  return G;
}

template <int F, int NT, bool BWD>
static void l12_launch_g(const L12Args& a, int G, unsigned blocks, hipStream_t st) {
#define TMGCN_L12(G_)                                                                                       \
  if (BWD) hipLaunchKernelGGL((l12_bwd_kernel<2, F, NT, G_>), dim3(blocks), dim3(256), 0, st, a);              \
  else hipLaunchKernelGGL((l12_fwd_kernel<2, F, NT, G_>), dim3(blocks), dim3(256), 0, st, a);
  switch (G) {
    case 1: TMGCN_L12(1) break;
    case 2: TMGCN_L12(2) break;
    case 4: TMGCN_L12(4) break;
    case 8: TMGCN_L12(8) break;
    default: TMGCN_L12(16)
  }
#undef TMGCN_L12
}

template <bool BWD>
static void l12_launch(const L12Args& a, int F, int NT, int G, unsigned blocks, hipStream_t st) {
#define TMGCN_L12_N(F_)                                                \
  switch (NT) {                                                        \
    case 2: l12_launch_g<F_, 2, BWD>(a, G, blocks, st); break;          \
    case 4: l12_launch_g<F_, 4, BWD>(a, G, blocks, st); break;          \
    case 6: l12_launch_g<F_, 6, BWD>(a, G, blocks, st); break;          \
    default: l12_launch_g<F_, 8, BWD>(a, G, blocks, st);                \
  }
  switch (F) {
    case 2: TMGCN_L12_N(2) break;
    case 4: TMGCN_L12_N(4) break;
    case 6: TMGCN_L12_N(6) break;
    default: TMGCN_L12_N(8)
  }
#undef TMGCN_L12_N
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_layer12_supported(int32_t K0, int32_t F, int32_t Nf) {
  return (K0 == 2 && F >= 2 && F <= 8 && F % 2 == 0 && Nf >= 2 && Nf <= 8 && Nf % 2 == 0) ? 1 : 0;
}

extern "C" int tmgcn_layer12_fwd_f32(const int64_t* rowptr, const int32_t* col, const float* val, const float* H,
                                      const float* W1, int32_t act1, const float* W2, int32_t act2, int64_t n_rows,
                                      int32_t N, int32_t K0, int32_t F, int32_t Nf, float* Z, float* AX, float* pre2,
                                      float avg_nnz_per_row, void* stream) {
  TMGCN_REQUIRE(tmgcn_layer12_supported(K0, F, Nf), "layer12: unsupported widths %d -> %d -> %d (2 -> even <= 8 -> even <= 8)", K0, F, Nf);
  TMGCN_REQUIRE(n_rows >= 0 && N > 0 && n_rows % N == 0, "layer12: bad shape n_rows=%lld N=%d", (long long)n_rows, N);
  TMGCN_REQUIRE(act1 >= TMGCN_ACT_NONE && act1 <= TMGCN_ACT_SELU && act2 >= TMGCN_ACT_NONE && act2 <= TMGCN_ACT_SELU,
                "layer12: unknown activation");
  if (n_rows == 0) return TMGCN_OK;
  TMGCN_REQUIRE(rowptr && H && W1 && W2 && Z, "layer12: null pointer");
  TMGCN_REQUIRE(reinterpret_cast<uintptr_t>(H) % 8 == 0, "layer12: H must be 8-byte aligned");
  L12Args a{rowptr, col, val, H, W1, W2, nullptr, nullptr, Z, AX, pre2, nullptr, nullptr, nullptr, n_rows, N, act1, act2};
  const int G = l12_lanes(avg_nnz_per_row);
  const unsigned blocks = (unsigned)((n_rows * G + 255) / 256);
  l12_launch<false>(a, F, Nf, G, blocks, (hipStream_t)stream);
  return check_launch("layer12_fwd");
}

extern "C" int64_t tmgcn_layer12_bwd_workspace_bytes(int32_t K0, int32_t F) {
  return (int64_t)kL12MaxBlocks * K0 * F * (int64_t)sizeof(float);
}

extern "C" int tmgcn_layer12_bwd_f32(const int64_t* t_rowptr, const int32_t* t_col, const float* t_val, const float* dZ,
                                      const float* pre2, const float* H, const float* W1, int32_t act1, const float* W2,
                                      int32_t act2, int64_t n_rows, int32_t N, int32_t K0, int32_t F, int32_t Nf,
                                      float* dW1, float avg_nnz_per_row, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
  TMGCN_REQUIRE(tmgcn_layer12_supported(K0, F, Nf), "layer12_bwd: unsupported widths %d -> %d -> %d", K0, F, Nf);
  TMGCN_REQUIRE(n_rows > 0 && N > 0 && n_rows % N == 0, "layer12_bwd: bad shape n_rows=%lld N=%d", (long long)n_rows, N);
  TMGCN_REQUIRE(act1 >= TMGCN_ACT_NONE && act1 <= TMGCN_ACT_SELU && act2 >= TMGCN_ACT_NONE && act2 <= TMGCN_ACT_SELU,
                "layer12_bwd: unknown activation");
  TMGCN_REQUIRE(t_rowptr && dZ && H && W1 && W2 && dW1 && workspace, "layer12_bwd: null pointer");
  TMGCN_REQUIRE((act2 == TMGCN_ACT_NONE) == (pre2 == nullptr), "layer12_bwd: pre2 must be given exactly when act2 is not none");
  TMGCN_REQUIRE(reinterpret_cast<uintptr_t>(H) % 8 == 0 && reinterpret_cast<uintptr_t>(dZ) % 8 == 0 &&
                    (!pre2 || reinterpret_cast<uintptr_t>(pre2) % 8 == 0),
                "layer12_bwd: H, dZ and pre2 must be 8-byte aligned");
  if (workspace_bytes < tmgcn_layer12_bwd_workspace_bytes(K0, F)) {
    set_error("layer12_bwd: workspace too small");
    return TMGCN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  L12Args a{t_rowptr, t_col, t_val, H, W1, W2, dZ, pre2, nullptr, nullptr, nullptr, dW1, (float*)workspace,
            acquire_sync_word(st), n_rows, N, act1, act2};
  TMGCN_REQUIRE(a.sync, "layer12_bwd: no hand-off word");
  const int G = l12_lanes(avg_nnz_per_row);
  int64_t blocks = (n_rows * G + 255) / 256;
  if (blocks > kL12MaxBlocks) blocks = kL12MaxBlocks;
  l12_launch<true>(a, F, Nf, G, (unsigned)blocks, st);
  return check_launch("layer12_bwd");
}

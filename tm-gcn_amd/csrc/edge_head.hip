// P4 — edge head: out[e] = [Z[src[e]] , Z[dst[e]]] · U  and its backward   (gfx950 / CDNA4)
//
// Replaces  Y.reshape(-1,F)[edge_src_nodes], [edge_trg_nodes], t.cat(...,1), t.matmul(.,U)
// (embedding_help_functions.py:228-232, 351-355, 491-495) and autograd through them
// (index backward = scatter-add into dZ, dU = catᵀ·dout).
//
// Link-prediction configs label 20x the real edges (E ≈ 3.2 M at the Reddit-shaped size,
// F = 6, C = 2): as separate torch ops this is eight full passes over E-sized tensors plus an
// atomic scatter; here it is one gather-and-dot kernel forward, and backward two kernels that
// use no atomics:
//   dZ   per row r:  S_src[c] = Σ_{e: src[e]=r} dout[e][c],  S_dst likewise (inverted edge index,
//        built once per edge set), then  dZ[r][f] = Σ_c S_src[c]·U[f][c] + S_dst[c]·U[F+f][c]
//        — the sum over edges is taken BEFORE the product with U, so the work is E·C, not E·F.
//   dU   row-chunk slabs of  [Z[src[e]], Z[dst[e]]]ᵀ · dout[e]  with fp64 running sums, reduced
//        in fixed order.
// Everything is summed in a fixed order: bitwise reproducible (torch's index_put backward on
// the GPU uses float atomics and is not).
//
// Each of the three has a generic form (G lanes per edge / row, any F <= 256, C <= 8) and the
// specialisations the launcher picks from the widths, the average row degree and the pointer
// alignment: *_small<F,C> for the reference's own heads (even F <= 8, C <= 4: one edge per lane,
// everything in registers), for wide heads a two-kernel dZ (entry sums, then an in-place expand)
// and a lane-per-feature dU.  All forms produce the same bits (tools/ab_edge_head.py).  The index
// arrays are int64 (the reference's) or int32 (template parameter IT; the *_i32 entry points).
#include "common.h"

namespace tmgcn {

constexpr int kMaxC = 8;     // classes
constexpr int kMaxF = 256;   // embedding width handled by the fused head

// IT = index type of the edge arrays: int64_t (the reference's flat index, ehf:196-198) or int32_t
// (the *_i32 entry points: half the index bytes when rows and 2·E fit 31 bits)
template <typename IT>
struct EdgeArgs {
  const float* Z;       // [R][F]
  const IT* src;        // [E] flat row index t*N+node (ehf:196-198)
  const IT* dst;        // [E]
  const float* U;       // [2F][C]
  float* out;           // [E][C]
  int64_t E;
  int32_t F, C;
};

// G lanes share one edge: lane gl takes features gl, gl+G, ... of both endpoint rows (coalesced
// across the group), partial dot products are combined with a shuffle butterfly.  U sits in LDS.
template <int G, typename IT>
__global__ __launch_bounds__(256) void edge_head_fwd_kernel(EdgeArgs<IT> a) {
  extern __shared__ float Us[];  // [2F][C]
  for (int t = threadIdx.x; t < 2 * a.F * a.C; t += 256) Us[t] = a.U[t];
  __syncthreads();
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t e = gid / G;
  const int gl = (int)(gid % G);
  const bool live = e < a.E;
  float acc[kMaxC];
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) acc[c] = 0.f;
  if (live) {
    const float* zs = a.Z + (int64_t)a.src[e] * a.F;
    const float* zd = a.Z + (int64_t)a.dst[e] * a.F;
    // (feature quads per lane with 16-byte loads were tried: 1.3-3x slower — the U reads from LDS
    // then stride 4·C floats across lanes and bank-conflict)
    for (int f = gl; f < a.F; f += G) {
      const float s = zs[f], d = zd[f];
      const float* us = Us + f * a.C;
      const float* ud = Us + (a.F + f) * a.C;
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < a.C) acc[c] = fmaf(d, ud[c], fmaf(s, us[c], acc[c]));
    }
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) acc[c] += __shfl_xor(acc[c], o);
  if (live && gl == 0) {
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < a.C) a.out[e * a.C + c] = acc[c];
  }
}

// Narrow heads (the reference's: F = 6 or 2, C = 2): one edge per lane with both widths known at
// compile time — the 2·FT/2 row loads are 8-byte vectors all in flight at once, U is read through
// uniform (scalar) loads into SGPRs, the CT logits leave as one vector store.  Needs an 8-byte
// aligned Z (and out for CT = 2 / 4): checked by the launcher.
template <int FT, int CT, typename IT>
__global__ __launch_bounds__(256) void edge_head_fwd_small_kernel(EdgeArgs<IT> a) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= a.E) return;
  const float2* zs = reinterpret_cast<const float2*>(a.Z + (int64_t)a.src[e] * FT);
  const float2* zd = reinterpret_cast<const float2*>(a.Z + (int64_t)a.dst[e] * FT);
  float2 s[FT / 2], d[FT / 2];
#pragma unroll
  for (int i = 0; i < FT / 2; ++i) {
    s[i] = zs[i];
    d[i] = zd[i];
  }
  const float* __restrict__ U = a.U;
  float acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = 0.f;
#pragma unroll
  for (int i = 0; i < FT / 2; ++i)  // same order of accumulation as the generic kernel: f ascending, src then dst
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = 2 * i + h;
      const float sv = h ? s[i].y : s[i].x, dv = h ? d[i].y : d[i].x;
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[c] = fmaf(dv, U[(FT + f) * CT + c], fmaf(sv, U[f * CT + c], acc[c]));
    }
  float* o = a.out + e * CT;
  if constexpr (CT == 2) {
    *reinterpret_cast<float2*>(o) = make_float2(acc[0], acc[1]);
  } else if constexpr (CT == 4) {
    *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
#pragma unroll
    for (int c = 0; c < CT; ++c) o[c] = acc[c];
  }
}

template <typename IT>
struct EdgeBwdArgs {
  const float* Z;
  const IT* src;
  const IT* dst;
  const float* U;
  const float* dout;     // [E][C]
  const IT* eptr;        // [R+1] inverted index: entries of row r are eidx[eptr[r] .. eptr[r+1])
  const IT* eidx;        // [2E]  entry = 2*edge + role (0: the row is the edge's src, 1: its dst)
  float* dZ;             // [R][F]
  float* part;           // dU slabs [chunks][2F][C]
  int64_t R, E;
  int32_t F, C;
  int32_t chunks;
  int64_t edges_per_chunk;
  int32_t du_edges;      // edges per LDS tile of the dU kernel
};

// G lanes per row: the lanes split the row's incident entries (lane gl takes entries gl, gl+G, ...:
// G independent chains of dependent index -> dout loads instead of one), the partial sums are
// combined with an xor butterfly — every lane ends with the same bits, in an order fixed by G —
// then lane gl writes features gl, gl+G, ...
template <int G, typename IT>
__global__ __launch_bounds__(256) void edge_head_dz_kernel(EdgeBwdArgs<IT> a) {
  extern __shared__ float Us[];
  for (int t = threadIdx.x; t < 2 * a.F * a.C; t += 256) Us[t] = a.U[t];
  __syncthreads();
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = gid / G;
  const int gl = (int)(gid % G);
  if (r >= a.R) return;
  double S[2][kMaxC];
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) S[0][c] = S[1][c] = 0.0;
  const int64_t p_end = a.eptr[r + 1];
  for (int64_t p = a.eptr[r] + gl; p < p_end; p += G) {
    const int64_t x = a.eidx[p];
    const float* g = a.dout + (int64_t)(x >> 1) * a.C;
    const bool is_dst = x & 1;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < a.C) {
        const double v = (double)g[c];
        if (is_dst) S[1][c] += v; else S[0][c] += v;
      }
  }
  if (G > 1) {
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < a.C) {
          S[0][c] += __shfl_xor(S[0][c], o);
          S[1][c] += __shfl_xor(S[1][c], o);
        }
  }
  float s0[kMaxC], s1[kMaxC];
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) {
    s0[c] = (float)S[0][c];
    s1[c] = (float)S[1][c];
  }
  for (int f = gl; f < a.F; f += G) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < a.C) v = fmaf(s1[c], Us[(a.F + f) * a.C + c], fmaf(s0[c], Us[f * a.C + c], v));
    a.dZ[r * a.F + f] = v;
  }
}

// Wide heads (F >= 16): the two halves of the kernel above want different lane counts — the entry
// sums a few lanes per row (avg entries / 2), the product with U one lane per four features — so
// they run as two kernels.  edge_head_dz_sums leaves S_src[c], S_dst[c] (as floats: what the fused
// kernel multiplies too) in the first 2C floats of the row's dZ storage; edge_head_dz_expand reads
// them back and overwrites the row with  dZ[r][f] = Σ_c S_src[c]·U[f][c] + S_dst[c]·U[F+f][c]  as
// 16-byte stores.  All lanes of a row sit in one wave and load S before any of them stores (one
// instruction stream), so expanding in place is safe; same sums in the same order as the fused
// kernel, hence the same bits.
template <int G, typename IT>
__global__ __launch_bounds__(256) void edge_head_dz_sums_kernel(EdgeBwdArgs<IT> a) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = gid / G;
  const int gl = (int)(gid % G);
  if (r >= a.R) return;
  double S[2][kMaxC];
#pragma unroll
  for (int c = 0; c < kMaxC; ++c) S[0][c] = S[1][c] = 0.0;
  const int64_t p_end = a.eptr[r + 1];
  for (int64_t p = a.eptr[r] + gl; p < p_end; p += G) {
    const int64_t x = a.eidx[p];
    const float* g = a.dout + (int64_t)(x >> 1) * a.C;
    const bool is_dst = x & 1;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < a.C) {
        const double v = (double)g[c];
        if (is_dst) S[1][c] += v; else S[0][c] += v;
      }
  }
  if (G > 1) {
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < a.C) {
          S[0][c] += __shfl_xor(S[0][c], o);
          S[1][c] += __shfl_xor(S[1][c], o);
        }
  }
  if (gl == 0) {
    float* o = a.dZ + r * a.F;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < a.C) {
        o[c] = (float)S[0][c];
        o[a.C + c] = (float)S[1][c];
      }
  }
}

// LP lanes per row (a power of two >= F/4, <= 64), 64/LP rows per wave.  Persistent: a lane keeps its
// four feature columns of U (both roles) in registers and strides over the rows.
// CM = compile-time bound on the classes (2 or kMaxC): sizes the register arrays, i.e. the occupancy.
template <int CM, typename IT>
__global__ __launch_bounds__(256) void edge_head_dz_expand_kernel(EdgeBwdArgs<IT> a, int LP) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int fl = (int)(gid % LP);
  const bool live = 4 * fl < a.F;
  float us[4][CM], ud[4][CM];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < CM; ++c) {
      const int f = 4 * fl + i;
      us[i][c] = (live && c < a.C) ? a.U[f * a.C + c] : 0.f;
      ud[i][c] = (live && c < a.C) ? a.U[(a.F + f) * a.C + c] : 0.f;
    }
  const int64_t row_stride = (int64_t)gridDim.x * 256 / LP;
  constexpr int RU = 4;  // rows in flight per lane group: the S loads of all of them are requested first
  for (int64_t r = gid / LP; r < a.R; r += RU * row_stride) {
    float s0[RU][CM], s1[RU][CM];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t ru = r + u * row_stride;
      const float* row = a.dZ + (ru < a.R ? ru : r) * a.F;
#pragma unroll
      for (int c = 0; c < CM; ++c) {
        s0[u][c] = c < a.C ? row[c] : 0.f;
        s1[u][c] = c < a.C ? row[a.C + c] : 0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // every lane of a row's wave has read S before any store below
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t ru = r + u * row_stride;
      if (live && ru < a.R) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float t = 0.f;
#pragma unroll
          for (int c = 0; c < CM; ++c)
            if (c < a.C) t = fmaf(s1[u][c], ud[i][c], fmaf(s0[u][c], us[i][c], t));
          v[i] = t;
        }
        *reinterpret_cast<float4*>(a.dZ + ru * a.F + 4 * fl) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// dU slabs for wide heads (F >= 64): lane = feature.  Each wave owns a contiguous quarter of the
// block's edges and walks it in batches of DUW_B (8 for binary heads): lanes 0..DUW_B-1 fetch the batch's endpoints and
// upstream rows (coalesced), the endpoint rows of ALL edges of the batch are then requested back to
// back (wave-uniform base from v_readlane + lane offset: coalesced 256-byte row segments) before
// the first is used — two memory latencies per batch instead of two per edge.  Lane l holds
// features l, l+64, ... of both roles and keeps its own (feature, class) sums in fp64 registers; no
// LDS and no barrier inside the loop.  The four waves are folded in order through LDS at the end.
// NJ = feature slots per lane (ceil(F/64)), CM = compile-time bound on the classes (2 or CM):
// sizes the register arrays (NJ = 4, CM = 8 needs 312 VGPRs; the common 128-wide binary head 80)
// DUW_B = edges per batch
template <int NJ, int CM, int DUW_B, typename IT>
__global__ __launch_bounds__(256) void edge_head_du_wide_kernel(EdgeBwdArgs<IT> a) {
  extern __shared__ double redw[];  // [3][2F*C]: waves 1..3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_out = 2 * a.F * a.C;
  const int64_t c0 = (int64_t)blockIdx.x * a.edges_per_chunk;
  int64_t c1 = c0 + a.edges_per_chunk;
  if (c1 > a.E) c1 = a.E;
  const int64_t q = (c1 - c0 + 3) / 4;  // edges per wave
  const int64_t e0 = c0 + wave * q;
  int64_t e1 = e0 + q;
  if (e1 > c1) e1 = c1;
  double acc[NJ][2][CM];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int c = 0; c < CM; ++c) acc[j][0][c] = acc[j][1][c] = 0.0;
  for (int64_t e = e0; e < e1; e += DUW_B) {
    const int nb = (int)((e1 - e) < DUW_B ? (e1 - e) : DUW_B);
    // lanes 0..nb-1: this batch's endpoints (as row offsets in floats) and upstream rows
    int64_t so = 0, dof = 0;
    float gl[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) gl[c] = 0.f;
    if (lane < nb) {
      so = (int64_t)a.src[e + lane] * a.F;
      dof = (int64_t)a.dst[e + lane] * a.F;
#pragma unroll
      for (int c = 0; c < CM; ++c)
        if (c < a.C) gl[c] = a.dout[(e + lane) * a.C + c];
    }
    float s[DUW_B][NJ], d[DUW_B][NJ];
#pragma unroll
    for (int i = 0; i < DUW_B; ++i) {
      // lanes past the batch hold offset 0: a valid row, multiplied by an upstream row of zeros
      const int64_t sb = __shfl(so, i), db = __shfl(dof, i);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int f = lane + 64 * j;
        const bool ok = f < a.F;
        s[i][j] = ok ? a.Z[sb + f] : 0.f;
        d[i][j] = ok ? a.Z[db + f] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < DUW_B; ++i) {
      double g[CM];
#pragma unroll
      for (int c = 0; c < CM; ++c) g[c] = c < a.C ? (double)__shfl(gl[c], i) : 0.0;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int c = 0; c < CM; ++c)
          if (c < a.C) {
            acc[j][0][c] = fma((double)s[i][j], g[c], acc[j][0][c]);
            acc[j][1][c] = fma((double)d[i][j], g[c], acc[j][1][c]);
          }
    }
  }
  if (wave) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int role = 0; role < 2; ++role)
#pragma unroll
        for (int c = 0; c < CM; ++c) {
          const int f = lane + 64 * j;
          if (f < a.F && c < a.C) redw[(int64_t)(wave - 1) * n_out + (role * a.F + f) * a.C + c] = acc[j][role][c];
        }
  }
  __syncthreads();
  if (wave == 0) {
    float* P = a.part + (int64_t)blockIdx.x * n_out;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int role = 0; role < 2; ++role)
#pragma unroll
        for (int c = 0; c < CM; ++c) {
          const int f = lane + 64 * j;
          if (f < a.F && c < a.C) {
            const int o = (role * a.F + f) * a.C + c;
            P[o] = (float)(((acc[j][role][c] + redw[o]) + redw[n_out + o]) + redw[2 * n_out + o]);
          }
        }
  }
}

// dU slabs: edges staged through LDS in tiles of du_edges.  n_out = 2F*C outputs (k, c); with
// n_out <= 128 the threads form 256/n_out edge groups so every lane works (the real head is
// 12 x 2 = 24 outputs), otherwise each thread owns up to 16 outputs; fp64 running sums, groups
// combined through LDS in fixed order.
constexpr int DU_OMAX = 16;  // 2*256*8 / 256
template <typename IT>
__global__ __launch_bounds__(256) void edge_head_du_kernel(EdgeBwdArgs<IT> a) {
  extern __shared__ float sm[];  // [du_edges][2F] gathered rows, then [du_edges][C] dout rows
  __shared__ double red[256];
  const int K = 2 * a.F;
  float* sz = sm;
  float* sd = sm + a.du_edges * K;
  const int n_out = K * a.C;
  const int groups = n_out <= 128 ? 256 / n_out : 1;
  const int grp = groups > 1 ? threadIdx.x / n_out : 0;
  const int64_t e0 = (int64_t)blockIdx.x * a.edges_per_chunk;
  int64_t e1 = e0 + a.edges_per_chunk;
  if (e1 > a.E) e1 = a.E;
  double acc[DU_OMAX];
#pragma unroll
  for (int o = 0; o < DU_OMAX; ++o) acc[o] = 0.0;
  for (int64_t e = e0; e < e1; e += a.du_edges) {
    const int ne = (int)((e1 - e) < a.du_edges ? (e1 - e) : a.du_edges);
    __syncthreads();
    for (int t = threadIdx.x; t < ne * K; t += 256) {
      const int i = t / K, k = t % K;
      const int64_t row = k < a.F ? (int64_t)a.src[e + i] : (int64_t)a.dst[e + i];
      sz[t] = a.Z[row * a.F + (k < a.F ? k : k - a.F)];
    }
    for (int t = threadIdx.x; t < ne * a.C; t += 256) sd[t] = a.dout[e * a.C + t];
    __syncthreads();
    if (groups > 1) {
      if (grp < groups) {
        const int idx = threadIdx.x % n_out, k = idx / a.C, c = idx % a.C;
        double s = acc[0];
        for (int i = grp; i < ne; i += groups) s += (double)sz[i * K + k] * (double)sd[i * a.C + c];
        acc[0] = s;
      }
    } else {
#pragma unroll
      for (int o = 0; o < DU_OMAX; ++o) {
        const int idx = threadIdx.x + o * 256;
        if (idx < n_out) {
          const int k = idx / a.C, c = idx % a.C;
          double s = acc[o];
          for (int i = 0; i < ne; ++i) s += (double)sz[i * K + k] * (double)sd[i * a.C + c];
          acc[o] = s;
        }
      }
    }
  }
  float* P = a.part + (int64_t)blockIdx.x * n_out;
  if (groups > 1) {
    __syncthreads();
    red[threadIdx.x] = grp < groups ? acc[0] : 0.0;
    __syncthreads();
    if (threadIdx.x < n_out) {
      double s = 0.0;
      for (int g = 0; g < groups; ++g) s += red[g * n_out + threadIdx.x];
      P[threadIdx.x] = (float)s;
    }
  } else {
#pragma unroll
    for (int o = 0; o < DU_OMAX; ++o) {
      const int idx = threadIdx.x + o * 256;
      if (idx < n_out) P[idx] = (float)acc[o];
    }
  }
}

// dU slabs for narrow heads (2·FT·CT <= 64 outputs): every lane keeps ALL outputs of its own edges
// (edge e0 + t, e0 + t + 256, ... of the block's chunk) in fp64 registers — no LDS staging, no
// barrier inside the edge loop, the row / dout loads of successive edges are independent — and the
// block folds them once at the end: xor butterfly inside each wave, then the four waves in order.
template <int FT, int CT, typename IT>
__global__ __launch_bounds__(256) void edge_head_du_small_kernel(EdgeBwdArgs<IT> a) {
  constexpr int K = 2 * FT, NO = K * CT;
  __shared__ double red[4][NO];
  const int64_t e0 = (int64_t)blockIdx.x * a.edges_per_chunk;
  int64_t e1 = e0 + a.edges_per_chunk;
  if (e1 > a.E) e1 = a.E;
  double acc[K][CT];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[k][c] = 0.0;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
    const float2* zs = reinterpret_cast<const float2*>(a.Z + (int64_t)a.src[e] * FT);
    const float2* zd = reinterpret_cast<const float2*>(a.Z + (int64_t)a.dst[e] * FT);
    float z[K], g[CT];
#pragma unroll
    for (int i = 0; i < FT / 2; ++i) {
      const float2 sv = zs[i], dv = zd[i];
      z[2 * i] = sv.x;
      z[2 * i + 1] = sv.y;
      z[FT + 2 * i] = dv.x;
      z[FT + 2 * i + 1] = dv.y;
    }
    if constexpr (CT == 2) {
      const float2 gv = *reinterpret_cast<const float2*>(a.dout + e * CT);
      g[0] = gv.x;
      g[1] = gv.y;
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c) g[c] = a.dout[e * CT + c];
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[k][c] = fma((double)z[k], (double)g[c], acc[k][c]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      double v = acc[k][c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) red[wave][k * CT + c] = v;
    }
  __syncthreads();
  if (threadIdx.x < NO)
    a.part[(int64_t)blockIdx.x * NO + threadIdx.x] =
        (float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

// one block per output element: the 256 threads stride over the chunk slabs, xor butterfly per wave,
// the four waves in order (a fixed summation order)
__global__ __launch_bounds__(256) void edge_head_du_reduce_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ dU, int n_out, int chunks) {
  __shared__ double sh[4];
  const int o = blockIdx.x;
  double s = 0.0;
  for (int c = threadIdx.x; c < chunks; c += 256) s += (double)part[(int64_t)c * n_out + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dU[o] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

static int lanes_per_item(int F) {  // G: 1 for the real (tiny) heads, up to 64 lanes for wide ones
  if (F <= 8) return 1;
  int g = 2;
  while (g < 64 && g * 4 < F) g <<= 1;
  return g;
}

// narrow-head kernels: even F <= 8, C <= 4, 8-byte aligned operands
static bool small_head(int F, int C, const void* Z, const void* io) {
  return F >= 2 && F <= 8 && F % 2 == 0 && C >= 1 && C <= 4 && reinterpret_cast<uintptr_t>(Z) % 8 == 0 &&
         reinterpret_cast<uintptr_t>(io) % 16 == 0;
}
#define TMGCN_HEAD_C(KERNEL, FT, ...)                                                           \
  switch (C) {                                                                                  \
    case 1: hipLaunchKernelGGL((KERNEL<FT, 1, IT>), __VA_ARGS__); break;                            \
    case 2: hipLaunchKernelGGL((KERNEL<FT, 2, IT>), __VA_ARGS__); break;                            \
    case 3: hipLaunchKernelGGL((KERNEL<FT, 3, IT>), __VA_ARGS__); break;                            \
    default: hipLaunchKernelGGL((KERNEL<FT, 4, IT>), __VA_ARGS__);                                  \
  }
#define TMGCN_HEAD_FC(KERNEL, ...)                                                              \
  switch (F) {                                                                                  \
    case 2: TMGCN_HEAD_C(KERNEL, 2, __VA_ARGS__) break;                                         \
    case 4: TMGCN_HEAD_C(KERNEL, 4, __VA_ARGS__) break;                                         \
    case 6: TMGCN_HEAD_C(KERNEL, 6, __VA_ARGS__) break;                                         \
    default: TMGCN_HEAD_C(KERNEL, 8, __VA_ARGS__)                                               \
  }

#ifndef TMGCN_DZ_CHAIN
#define TMGCN_DZ_CHAIN 2   // entries per lane aimed at (A/B: tools/ab_edge_head.py, profiles/archive/r02x_ab_edge_head.txt)
#endif
#ifndef TMGCN_DZ_MAXG
#define TMGCN_DZ_MAXG 32
#endif
// lanes per row of the dZ kernel: at least the feature lanes, and enough to split a row's incident
// entries (avg = 2E/R) into chains of about two
static int dz_lanes(int F, int64_t E, int64_t R) {
  int g = F <= 8 ? 1 : 2;
  while (F > 8 && g < 64 && g * 4 < F) g <<= 1;
  const int64_t avg = R > 0 ? (2 * E + R - 1) / R : 0;
  while (g < TMGCN_DZ_MAXG && (int64_t)g * TMGCN_DZ_CHAIN < avg) g <<= 1;
  return g;
}

static int du_tile_edges(int F) {  // LDS tile <= 32 KB of gathered rows
  int e = 8192 / (2 * F);
  if (e > 64) e = 64;
  if (e < 4) e = 4;
  return e;
}

static void du_plan(int64_t E, int F, int* chunks, int64_t* per) {
  const int DU_EDGES = du_tile_edges(F);
  int64_t c = (E + 255) / 256;  // >= 256 edges per chunk, <= 2048 slabs
  if (c > 2048) c = 2048;
  if (c < 1) c = 1;
  int64_t p = (E + c - 1) / c;
  p = (p + DU_EDGES - 1) / DU_EDGES * DU_EDGES;
  c = p ? (E + p - 1) / p : 1;
  if (c < 1) c = 1;
  *chunks = (int)c;
  *per = p;
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_edge_head_supported(int32_t F, int32_t C) {
  return (F >= 1 && F <= kMaxF && C >= 1 && C <= kMaxC) ? 1 : 0;
}

template <typename IT>
static int edge_head_fwd_impl(const float* Z, const IT* src, const IT* dst, const float* U, float* out, int64_t E,
                              int32_t F, int32_t C, void* stream) {
  TMGCN_REQUIRE(tmgcn_edge_head_supported(F, C), "edge_head: unsupported widths F=%d C=%d (F <= %d, C <= %d)", F, C,
                kMaxF, kMaxC);
  TMGCN_REQUIRE(E >= 0, "edge_head: negative E");
  if (E == 0) return TMGCN_OK;
  TMGCN_REQUIRE(Z && src && dst && U && out, "edge_head: null pointer");
  EdgeArgs<IT> a{Z, src, dst, U, out, E, F, C};
  const size_t smem = (size_t)2 * F * C * sizeof(float);
  const int G = lanes_per_item(F);
  const unsigned grid = (unsigned)((E * G + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (small_head(F, C, Z, out)) {
    TMGCN_HEAD_FC(edge_head_fwd_small_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, a)
    return check_launch("edge_head_fwd_small");
  }
  switch (G) {
    case 1: hipLaunchKernelGGL((edge_head_fwd_kernel<1, IT>), dim3(grid), dim3(256), smem, st, a); break;
    case 2: hipLaunchKernelGGL((edge_head_fwd_kernel<2, IT>), dim3(grid), dim3(256), smem, st, a); break;
    case 4: hipLaunchKernelGGL((edge_head_fwd_kernel<4, IT>), dim3(grid), dim3(256), smem, st, a); break;
    case 8: hipLaunchKernelGGL((edge_head_fwd_kernel<8, IT>), dim3(grid), dim3(256), smem, st, a); break;
    case 16: hipLaunchKernelGGL((edge_head_fwd_kernel<16, IT>), dim3(grid), dim3(256), smem, st, a); break;
    case 32: hipLaunchKernelGGL((edge_head_fwd_kernel<32, IT>), dim3(grid), dim3(256), smem, st, a); break;
    default: hipLaunchKernelGGL((edge_head_fwd_kernel<64, IT>), dim3(grid), dim3(256), smem, st, a);
  }
  return check_launch("edge_head_fwd");
}

extern "C" int64_t tmgcn_edge_head_bwd_workspace_bytes(int64_t E, int32_t F, int32_t C) {
  if (E <= 0 || F <= 0 || C <= 0) return 0;
  int chunks;
  int64_t per;
  du_plan(E, F, &chunks, &per);
  return (int64_t)chunks * 2 * F * C * (int64_t)sizeof(float);
}

template <typename IT>
static int edge_head_bwd_impl(const float* Z, const IT* src, const IT* dst, const float* U, const float* dout,
                              const IT* eptr, const IT* eidx, float* dZ, float* dU, int64_t R, int64_t E,
                              int32_t F, int32_t C, void* workspace, int64_t workspace_bytes, void* stream) {
  TMGCN_REQUIRE(tmgcn_edge_head_supported(F, C), "edge_head_bwd: unsupported widths F=%d C=%d", F, C);
  TMGCN_REQUIRE(R >= 0 && E >= 0, "edge_head_bwd: negative extent");
  hipStream_t st = (hipStream_t)stream;
  int chunks;
  int64_t per;
  du_plan(E, F, &chunks, &per);
  EdgeBwdArgs<IT> a{Z, src, dst, U, dout, eptr, eidx, dZ, (float*)workspace, R, E, F, C, chunks, per, du_tile_edges(F)};
  if (dZ && R > 0) {
    TMGCN_REQUIRE(eptr && (E == 0 || (eidx && dout)) && U, "edge_head_bwd: null pointer (dZ)");
    const size_t smem = (size_t)2 * F * C * sizeof(float);
    if (F >= 16 && F % 4 == 0 && 2 * C <= F && reinterpret_cast<uintptr_t>(dZ) % 16 == 0) {
      // wide head: entry sums into the row's own storage, then the in-place product with U
      const int G = dz_lanes(1, E, R);
      const unsigned gs = (unsigned)((R * G + 255) / 256);
      switch (G) {
        case 1: hipLaunchKernelGGL((edge_head_dz_sums_kernel<1, IT>), dim3(gs), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((edge_head_dz_sums_kernel<2, IT>), dim3(gs), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((edge_head_dz_sums_kernel<4, IT>), dim3(gs), dim3(256), 0, st, a); break;
        case 8: hipLaunchKernelGGL((edge_head_dz_sums_kernel<8, IT>), dim3(gs), dim3(256), 0, st, a); break;
        case 16: hipLaunchKernelGGL((edge_head_dz_sums_kernel<16, IT>), dim3(gs), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((edge_head_dz_sums_kernel<32, IT>), dim3(gs), dim3(256), 0, st, a);
      }
      int rc = check_launch("edge_head_dz_sums");
      if (rc) return rc;
      int LP = 4;
      while (LP * 4 < F) LP <<= 1;
      int64_t gx = (R * LP + 255) / 256;  // 256 threads are a whole number of rows
      if (C <= 2) {
        const int64_t cap = persistent_grid(edge_head_dz_expand_kernel<2, IT>, 256) * 2;  // the helper caps at 4 blocks per CU
        hipLaunchKernelGGL((edge_head_dz_expand_kernel<2, IT>), dim3((unsigned)(gx < cap ? gx : cap)), dim3(256), 0, st, a, LP);
      } else {
        const int64_t cap = persistent_grid(edge_head_dz_expand_kernel<kMaxC, IT>, 256);
        hipLaunchKernelGGL((edge_head_dz_expand_kernel<kMaxC, IT>), dim3((unsigned)(gx < cap ? gx : cap)), dim3(256), 0, st, a, LP);
      }
      rc = check_launch("edge_head_dz_expand");
      if (rc) return rc;
    } else {
    const int G = dz_lanes(F, E, R);
    const unsigned grid = (unsigned)((R * G + 255) / 256);
    switch (G) {
      case 1: hipLaunchKernelGGL((edge_head_dz_kernel<1, IT>), dim3(grid), dim3(256), smem, st, a); break;
      case 2: hipLaunchKernelGGL((edge_head_dz_kernel<2, IT>), dim3(grid), dim3(256), smem, st, a); break;
      case 4: hipLaunchKernelGGL((edge_head_dz_kernel<4, IT>), dim3(grid), dim3(256), smem, st, a); break;
      case 8: hipLaunchKernelGGL((edge_head_dz_kernel<8, IT>), dim3(grid), dim3(256), smem, st, a); break;
      case 16: hipLaunchKernelGGL((edge_head_dz_kernel<16, IT>), dim3(grid), dim3(256), smem, st, a); break;
      case 32: hipLaunchKernelGGL((edge_head_dz_kernel<32, IT>), dim3(grid), dim3(256), smem, st, a); break;
      default: hipLaunchKernelGGL((edge_head_dz_kernel<64, IT>), dim3(grid), dim3(256), smem, st, a);
    }
    int rc = check_launch("edge_head_dz");
    if (rc) return rc;
    }
  }
  if (dU) {
    if (E == 0) {
      (void)hipMemsetAsync(dU, 0, (size_t)2 * F * C * sizeof(float), st);
      return check_launch("edge_head_du memset");
    }
    TMGCN_REQUIRE(Z && src && dst && dout, "edge_head_bwd: null pointer (dU)");
    const int64_t need = (int64_t)chunks * 2 * F * C * (int64_t)sizeof(float);
    if (!workspace || workspace_bytes < need) {
      set_error("edge_head_bwd: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
      return TMGCN_ERR_WORKSPACE;
    }
    if (small_head(F, C, Z, dout)) {
      TMGCN_HEAD_FC(edge_head_du_small_kernel, dim3((unsigned)chunks), dim3(256), 0, st, a)
    } else if (F >= 64) {
      const dim3 g((unsigned)chunks), b(256);
      const size_t sm = (size_t)3 * 2 * F * C * sizeof(double);
      const int nj = (F + 63) / 64;
#define TMGCN_DUW(NJ_)                                                                              \
  if (C <= 2) hipLaunchKernelGGL((edge_head_du_wide_kernel<NJ_, 2, 8, IT>), g, b, sm, st, a);           \
  else hipLaunchKernelGGL((edge_head_du_wide_kernel<NJ_, kMaxC, (NJ_ <= 2 ? 4 : 2), IT>), g, b, sm, st, a);
      switch (nj) {
        case 1: TMGCN_DUW(1) break;
        case 2: TMGCN_DUW(2) break;
        case 3: TMGCN_DUW(3) break;
        default: TMGCN_DUW(4)
      }
#undef TMGCN_DUW
    } else {
      const size_t smem = (size_t)a.du_edges * (2 * F + C) * sizeof(float);
      hipLaunchKernelGGL(edge_head_du_kernel<IT>, dim3((unsigned)chunks), dim3(256), smem, st, a);
    }
    int rc = check_launch("edge_head_du");
    if (rc) return rc;
    const int n_out = 2 * F * C;
    hipLaunchKernelGGL(edge_head_du_reduce_kernel, dim3(n_out), dim3(256), 0, st,
                       (const float*)workspace, dU, n_out, chunks);
    return check_launch("edge_head_du_reduce");
  }
  return TMGCN_OK;
}

extern "C" int tmgcn_edge_head_fwd_f32(const float* Z, const int64_t* src, const int64_t* dst,
                                        const float* U, float* out, int64_t E, int32_t F, int32_t C,
                                        void* stream) {
  return edge_head_fwd_impl<int64_t>(Z, src, dst, U, out, E, F, C, stream);
}

extern "C" int tmgcn_edge_head_fwd_i32_f32(const float* Z, const int32_t* src, const int32_t* dst,
                                            const float* U, float* out, int64_t E, int32_t F, int32_t C,
                                            void* stream) {
  return edge_head_fwd_impl<int32_t>(Z, src, dst, U, out, E, F, C, stream);
}

extern "C" int tmgcn_edge_head_bwd_f32(const float* Z, const int64_t* src, const int64_t* dst,
                                        const float* U, const float* dout, const int64_t* eptr,
                                        const int64_t* eidx, float* dZ, float* dU, int64_t R, int64_t E,
                                        int32_t F, int32_t C, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  return edge_head_bwd_impl<int64_t>(Z, src, dst, U, dout, eptr, eidx, dZ, dU, R, E, F, C, workspace, workspace_bytes,
                                     stream);
}

extern "C" int tmgcn_edge_head_bwd_i32_f32(const float* Z, const int32_t* src, const int32_t* dst,
                                            const float* U, const float* dout, const int32_t* eptr,
                                            const int32_t* eidx, float* dZ, float* dU, int64_t R, int64_t E,
                                            int32_t F, int32_t C, void* workspace, int64_t workspace_bytes,
                                            void* stream) {
  TMGCN_REQUIRE(R < (int64_t)0x7fffffff && 2 * E < (int64_t)0x7fffffff, "edge_head_bwd_i32: R=%lld or 2E=%lld does not fit 31 bits",
                (long long)R, (long long)(2 * E));
  return edge_head_bwd_impl<int32_t>(Z, src, dst, U, dout, eptr, eidx, dZ, dU, R, E, F, C, workspace, workspace_bytes,
                                     stream);
}

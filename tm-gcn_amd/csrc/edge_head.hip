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
#include "common.h"

namespace tmgcn {

constexpr int kMaxC = 8;     // classes
constexpr int kMaxF = 256;   // embedding width handled by the fused head

struct EdgeArgs {
  const float* Z;       // [R][F]
  const int64_t* src;   // [E] flat row index t*N+node (ehf:196-198)
  const int64_t* dst;   // [E]
  const float* U;       // [2F][C]
  float* out;           // [E][C]
  int64_t E;
  int32_t F, C;
};

// G lanes share one edge: lane gl takes features gl, gl+G, ... of both endpoint rows (coalesced
// across the group), partial dot products are combined with a shuffle butterfly.  U sits in LDS.
template <int G>
__global__ __launch_bounds__(256) void edge_head_fwd_kernel(EdgeArgs a) {
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
    const float* zs = a.Z + a.src[e] * a.F;
    const float* zd = a.Z + a.dst[e] * a.F;
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

struct EdgeBwdArgs {
  const float* Z;
  const int64_t* src;
  const int64_t* dst;
  const float* U;
  const float* dout;     // [E][C]
  const int64_t* eptr;   // [R+1] inverted index: entries of row r are eidx[eptr[r] .. eptr[r+1])
  const int64_t* eidx;   // [2E]  entry = 2*edge + role (0: the row is the edge's src, 1: its dst)
  float* dZ;             // [R][F]
  float* part;           // dU slabs [chunks][2F][C]
  int64_t R, E;
  int32_t F, C;
  int32_t chunks;
  int64_t edges_per_chunk;
  int32_t du_edges;      // edges per LDS tile of the dU kernel
};

// G lanes per row: every lane of the group sums the row's incident dout rows (broadcast loads),
// then lane gl writes features gl, gl+G, ...
template <int G>
__global__ __launch_bounds__(256) void edge_head_dz_kernel(EdgeBwdArgs a) {
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
  for (int64_t p = a.eptr[r]; p < a.eptr[r + 1]; ++p) {
    const int64_t x = a.eidx[p];
    const float* g = a.dout + (x >> 1) * a.C;
    const bool is_dst = x & 1;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < a.C) {
        const double v = (double)g[c];
        if (is_dst) S[1][c] += v; else S[0][c] += v;
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

// dU slabs: edges staged through LDS in tiles of du_edges.  n_out = 2F*C outputs (k, c); with
// n_out <= 128 the threads form 256/n_out edge groups so every lane works (the real head is
// 12 x 2 = 24 outputs), otherwise each thread owns up to 16 outputs; fp64 running sums, groups
// combined through LDS in fixed order.
constexpr int DU_OMAX = 16;  // 2*256*8 / 256
__global__ __launch_bounds__(256) void edge_head_du_kernel(EdgeBwdArgs a) {
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
      const int64_t row = k < a.F ? a.src[e + i] : a.dst[e + i];
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

// one wave per output element: lanes stride over the chunk slabs, fixed butterfly
__global__ __launch_bounds__(256) void edge_head_du_reduce_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ dU, int n_out, int chunks) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= n_out) return;  // whole wave
  double s = 0.0;
  for (int c = lane; c < chunks; c += kWave) s += (double)part[(int64_t)c * n_out + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) dU[o] = (float)s;
}

static int lanes_per_item(int F) {  // G: 1 for the real (tiny) heads, up to 64 lanes for wide ones
  if (F <= 8) return 1;
  int g = 2;
  while (g < 64 && g * 4 < F) g <<= 1;
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

extern "C" int tmgcn_edge_head_fwd_f32(const float* Z, const int64_t* src, const int64_t* dst,
                                        const float* U, float* out, int64_t E, int32_t F, int32_t C,
                                        void* stream) {
  TMGCN_REQUIRE(tmgcn_edge_head_supported(F, C), "edge_head: unsupported widths F=%d C=%d (F <= %d, C <= %d)", F, C,
                kMaxF, kMaxC);
  TMGCN_REQUIRE(E >= 0, "edge_head: negative E");
  if (E == 0) return TMGCN_OK;
  TMGCN_REQUIRE(Z && src && dst && U && out, "edge_head: null pointer");
  EdgeArgs a{Z, src, dst, U, out, E, F, C};
  const size_t smem = (size_t)2 * F * C * sizeof(float);
  const int G = lanes_per_item(F);
  const unsigned grid = (unsigned)((E * G + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  switch (G) {
    case 1: hipLaunchKernelGGL(edge_head_fwd_kernel<1>, dim3(grid), dim3(256), smem, st, a); break;
    case 2: hipLaunchKernelGGL(edge_head_fwd_kernel<2>, dim3(grid), dim3(256), smem, st, a); break;
    case 4: hipLaunchKernelGGL(edge_head_fwd_kernel<4>, dim3(grid), dim3(256), smem, st, a); break;
    case 8: hipLaunchKernelGGL(edge_head_fwd_kernel<8>, dim3(grid), dim3(256), smem, st, a); break;
    case 16: hipLaunchKernelGGL(edge_head_fwd_kernel<16>, dim3(grid), dim3(256), smem, st, a); break;
    case 32: hipLaunchKernelGGL(edge_head_fwd_kernel<32>, dim3(grid), dim3(256), smem, st, a); break;
    default: hipLaunchKernelGGL(edge_head_fwd_kernel<64>, dim3(grid), dim3(256), smem, st, a);
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

extern "C" int tmgcn_edge_head_bwd_f32(const float* Z, const int64_t* src, const int64_t* dst,
                                        const float* U, const float* dout, const int64_t* eptr,
                                        const int64_t* eidx, float* dZ, float* dU, int64_t R, int64_t E,
                                        int32_t F, int32_t C, void* workspace, int64_t workspace_bytes,
                                        void* stream) {
  TMGCN_REQUIRE(tmgcn_edge_head_supported(F, C), "edge_head_bwd: unsupported widths F=%d C=%d", F, C);
  TMGCN_REQUIRE(R >= 0 && E >= 0, "edge_head_bwd: negative extent");
  hipStream_t st = (hipStream_t)stream;
  int chunks;
  int64_t per;
  du_plan(E, F, &chunks, &per);
  EdgeBwdArgs a{Z, src, dst, U, dout, eptr, eidx, dZ, (float*)workspace, R, E, F, C, chunks, per, du_tile_edges(F)};
  if (dZ && R > 0) {
    TMGCN_REQUIRE(eptr && (E == 0 || (eidx && dout)) && U, "edge_head_bwd: null pointer (dZ)");
    const size_t smem = (size_t)2 * F * C * sizeof(float);
    const int G = lanes_per_item(F);
    const unsigned grid = (unsigned)((R * G + 255) / 256);
    switch (G) {
      case 1: hipLaunchKernelGGL(edge_head_dz_kernel<1>, dim3(grid), dim3(256), smem, st, a); break;
      case 2: hipLaunchKernelGGL(edge_head_dz_kernel<2>, dim3(grid), dim3(256), smem, st, a); break;
      case 4: hipLaunchKernelGGL(edge_head_dz_kernel<4>, dim3(grid), dim3(256), smem, st, a); break;
      case 8: hipLaunchKernelGGL(edge_head_dz_kernel<8>, dim3(grid), dim3(256), smem, st, a); break;
      case 16: hipLaunchKernelGGL(edge_head_dz_kernel<16>, dim3(grid), dim3(256), smem, st, a); break;
      case 32: hipLaunchKernelGGL(edge_head_dz_kernel<32>, dim3(grid), dim3(256), smem, st, a); break;
      default: hipLaunchKernelGGL(edge_head_dz_kernel<64>, dim3(grid), dim3(256), smem, st, a);
    }
    int rc = check_launch("edge_head_dz");
    if (rc) return rc;
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
    const size_t smem = (size_t)a.du_edges * (2 * F + C) * sizeof(float);
    hipLaunchKernelGGL(edge_head_du_kernel, dim3((unsigned)chunks), dim3(256), smem, st, a);
    int rc = check_launch("edge_head_du");
    if (rc) return rc;
    const int n_out = 2 * F * C;
    hipLaunchKernelGGL(edge_head_du_reduce_kernel, dim3((n_out + 3) / 4), dim3(256), 0, st,
                       (const float*)workspace, dU, n_out, chunks);
    return check_launch("edge_head_du_reduce");
  }
  return TMGCN_OK;
}

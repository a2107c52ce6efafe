// P2 — batched CSR SpMM  Y[r] = sum_p val[p] * X[slice(r)*N + col[p]]   (gfx950 / CDNA4)
//
// Replaces the per-slice loop of t.sparse.mm calls in the reference
// (embedding_help_functions.py:206-207, 303-304, 310-311, 471-472) and, fed the
// transposed CSR, autograd's sparse.mm backward.  One launch covers all T slices:
// the batched matrix is block diagonal, row r = k*N + i belongs to slice k = r / N.
//
// HBM-bound gather: per stored non-zero the kernel moves 8 B of (col,val) and one
// F*4-byte row of X; per row 8 B of rowptr and one F*4-byte output row.  Three
// kernels cover the F regimes of the reference's configs:
//   spmm_vec4   F % 4 == 0, F >= 16   lanes across F (float4 per lane; F > 256 as column chunks of 256), S = 64/LPR
//                                      non-zero streams per wave, DPP/shuffle combine
//   spmm_small  F in {1,2,3,4,6,8}     lanes across non-zeros (G lanes per row),
//                                      wavefront-shuffle segmented sum
//   spmm_generic any other F           lanes across F, scalar loads
// No atomics anywhere: every row is summed in a fixed order.
#include "common.h"
#include "spmm_row.h"

#ifndef TMGCN_SPMM_U
#define TMGCN_SPMM_U 4      // gathers in flight per lane (F >= 64)
#endif
#ifndef TMGCN_SPMM_US
#define TMGCN_SPMM_US 2     // ... on short tiles (spmm_row.h), as a multiple of the above
#endif

namespace tmgcn {

// ---------------------------------------------------------------------------------
// vec4 kernel.  LPR = lanes per non-zero stream (power of two, >= F/4), S = 64/LPR
// streams per wave.  A wave owns one row at a time; the row's (col,val) pairs are
// fetched 64 at a time with one coalesced load per array and handed to the streams
// with ds_bpermute (__shfl); each stream gathers whole X rows as 16-B loads, U of
// them in flight per lane.  Rows of more than kLongRow entries are shared by the block's four
// waves, and the heaviest tiles are taken first (spmm_row.h).
// ---------------------------------------------------------------------------------
template <int LPR, int U, int US>       // US: gathers in flight per lane on short tiles
__global__ __launch_bounds__(256) void spmm_vec4_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float4* __restrict__ X, float4* __restrict__ Y,
    int64_t n_rows, int32_t N, int32_t F4, unsigned int* tile_counter, GiantPlan giant) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __shared__ unsigned int s_tile;
  __shared__ float4 s_part[4 * LPR];      // partial sums of a long row, one per wave (spmm_row.h)
  const TileMap tm = make_tile_map(n_rows, N);          // tiles restart at every slice (spmm_row.h)
  const int64_t n_tiles = tm.n_tiles;
  HeavyScan heavy;
  heavy.init(rowptr, tm);
  for (;;) {
    // the heavy tiles first (spmm_row.h: windows drawn from counter[1]); then persistent blocks draw 64-row tiles from
    // counter[0] (ascending: resident blocks stay inside one slice of X); see common.h
    int64_t tile = -1;
    if (heavy.scanning) tile = heavy.next(rowptr, tm, tile_counter + 1, &s_tile, lane);
    const bool scanning = heavy.scanning;
    if (!scanning) {
      __syncthreads();
      if (threadIdx.x == 0) s_tile = atomicAdd(tile_counter, 1u);
      __syncthreads();
      tile = s_tile;
      if (tile >= n_tiles) break;
    }
    int64_t slice0, r_begin, r_end;
    tile_extent(tm, tile, slice0, r_begin, r_end);
    if (r_begin + kTileRows < r_end) r_end = r_begin + kTileRows;
    TileRows rows;
    rows.load(rowptr, r_begin, r_end, lane);
    if (TMGCN_HEAVY_FIRST && !scanning && rows.entries > heavy.thr) continue;   // done in somebody's pass 1
    // a tile of few entries is walked entry-major, several rows per wave at once (spmm_row.h "Short tiles")
    const int n_tile_rows = (int)(r_end - r_begin);
    if (short_tile(rows, true)) {
      for (int c0 = 0; c0 < F4; c0 += kWave) {
        const int w4 = F4 - c0 < kWave ? F4 - c0 : kWave;
        gather_short_tile<LPR, US>(col, val, X + slice0 * (int64_t)N * F4 + c0, rows, n_tile_rows, w4, lane, wave, F4,
                                   [&](int rr, const float4& acc, int fl) {
                                     if (fl < w4) store_f4(&Y[(r_begin + rr) * F4 + c0 + fl], acc);
                                   });
      }
      continue;
    }
    // a row wider than 64 float4 (F > 256; LPR = 64 then) is gathered as column chunks of 256 floats: the row's entries are
    // walked once per chunk (col / val come from L1 the second time), every gather is still a contiguous 1 KB piece
    for (int c0 = 0; c0 < F4; c0 += kWave) {
      const int w4 = F4 - c0 < kWave ? F4 - c0 : kWave;     // == F4 whenever F <= 256
      for (int rr = wave; rr < (int)(r_end - r_begin); rr += 4) {
        if ((rows.long_mask >> rr) & 1) continue;
        const int64_t r = r_begin + rr;
        const float4 acc = gather_row<LPR, U>(col, val, X + slice0 * (int64_t)N * F4 + c0, readlane64(rows.beg, rr),
                                              readlane64(rows.end, rr), w4, lane, F4);
        if (lane < LPR && lane < w4) store_f4(&Y[r * F4 + c0 + lane], acc);
      }
      for (uint64_t m = rows.long_mask; m; m &= m - 1) {          // long rows: all four waves on each
        const int rr = __builtin_ctzll(m);
        const int64_t r = r_begin + rr;
        const int64_t slice = slice0;
        const int64_t beg = readlane64(rows.beg, rr), end = readlane64(rows.end, rr);
        float4 acc;
        const int gi = (giant.rows && end - beg > kGiantRow) ? giant_find(giant, r) : -1;   // block-uniform
        if (gi >= 0) {                                           // summed chunk by chunk in front of this launch
          if (wave != (rr & 3)) continue;
          acc = giant_row_sum(giant, gi, F4, lane, c0, w4);
        } else {
          acc = gather_long_row<LPR, U>(col, val, X + slice * (int64_t)N * F4 + c0, beg, end, w4, lane, wave, s_part, F4);
        }
        if (wave == (rr & 3) && lane < LPR && lane < w4) store_f4(&Y[r * F4 + c0 + lane], acc);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Giant-row pre-pass (spmm_row.h "Giant rows"): block c sums chunk c — kGiantChunk consecutive entries of one giant row —
// with the four-wave gather and stores the partial sum to ws[c][F].
// ---------------------------------------------------------------------------------
template <int LPR, int U>
__global__ __launch_bounds__(256) void spmm_giant_partial_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const float4* __restrict__ X, int32_t N, int32_t F4, const int64_t* __restrict__ g_rows,
    const int32_t* __restrict__ g_chunk_ptr, const int32_t* __restrict__ g_chunk_giant, float4* __restrict__ ws) {
  __shared__ float4 s_part[4 * LPR];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x;
  const int g = g_chunk_giant[c];
  const int64_t r = g_rows[g];
  int64_t beg = rowptr[r] + (int64_t)(c - g_chunk_ptr[g]) * kGiantChunk;
  int64_t end = rowptr[r + 1];
  if (beg > end) beg = end;
  if (end - beg > kGiantChunk) end = beg + kGiantChunk;
  const int64_t slice = r / N;
  for (int c0 = 0; c0 < F4; c0 += kWave) {
    const int w4 = F4 - c0 < kWave ? F4 - c0 : kWave;
    const float4 acc = gather_long_row<LPR, U>(col, val, X + slice * (int64_t)N * F4 + c0, beg, end, w4, lane, wave, s_part, F4);
    if (wave == 0 && lane < LPR && lane < w4) ws[(int64_t)c * F4 + c0 + lane] = acc;
  }
}

int launch_giant_partial(const char* who, const int64_t* rowptr, const int32_t* col, const float* val, const float* X, int32_t N,
                         int32_t F, const int64_t* giant_rows, const int32_t* giant_chunks, int32_t n_giant, int32_t n_giant_chunks,
                         float* ws, int64_t ws_bytes, hipStream_t st) {
  TMGCN_REQUIRE(n_giant >= 0 && n_giant_chunks >= 0, "%s: negative giant-row counts", who);
  if (n_giant == 0) return TMGCN_OK;
  TMGCN_REQUIRE(giant_rows && giant_chunks && ws && n_giant_chunks >= n_giant, "%s: giant-row plan: null pointer or fewer chunks than rows", who);
  TMGCN_REQUIRE(ws_bytes >= tmgcn_spmm_giant_workspace_bytes(n_giant_chunks, F), "%s: giant-row workspace too small", who);
  TMGCN_REQUIRE(F % 4 == 0 && F >= 16 && reinterpret_cast<uintptr_t>(ws) % 16 == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0,
                "%s: the giant-row plan needs F %% 4 == 0, F >= 16 and 16-byte aligned X / workspace", who);
  const int F4 = F / 4;
  int lpr = 4;
  while (lpr < F4 && lpr < 64) lpr <<= 1;
  const int32_t* chunk_ptr = giant_chunks;
  const int32_t* chunk_giant = giant_chunks + n_giant + 1;
  const float4* X4 = reinterpret_cast<const float4*>(X);
  float4* ws4 = reinterpret_cast<float4*>(ws);
#define TMGCN_GIANT_CASE(L, UU)                                                                                          \
  case L:                                                                                                                \
    hipLaunchKernelGGL((spmm_giant_partial_kernel<L, UU>), dim3((unsigned)n_giant_chunks), dim3(256), 0, st, rowptr, col, \
                       val, X4, N, F4, giant_rows, chunk_ptr, chunk_giant, ws4);                                         \
    break;
  switch (lpr) {     // the (LPR, U) pairs of spmm_vec4_kernel and of spmm_gemm_kernel: one row-sum order everywhere
    TMGCN_GIANT_CASE(4, 2)
    TMGCN_GIANT_CASE(8, 2)
    TMGCN_GIANT_CASE(16, TMGCN_SPMM_U)
    TMGCN_GIANT_CASE(32, TMGCN_SPMM_U)
    TMGCN_GIANT_CASE(64, TMGCN_SPMM_U)
  }
#undef TMGCN_GIANT_CASE
  return check_launch(who);
}

// ---------------------------------------------------------------------------------
// small-F kernel: G lanes share one row and stride over its non-zeros; the F partial
// sums per lane are combined with a wavefront-shuffle butterfly (segmented sum).
// ---------------------------------------------------------------------------------
template <int F, int G>
__global__ __launch_bounds__(256) void spmm_small_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ X, float* __restrict__ Y,
    int64_t n_rows, int32_t N) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = gid / G;
  const int gl = (int)(gid % G);
  const bool live = r < n_rows;
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  int64_t beg = 0, end = 0;
  if (live) {
    beg = rowptr[r];
    end = rowptr[r + 1];
  }
  // rows that would take this group more than kNarrowLong trips are left to the whole wave below (spmm_row.h)
  const bool is_long = G < kWave && end - beg > (int64_t)kNarrowLong * G;
  if (live && !is_long) {
    const int64_t xoff = (r / N) * (int64_t)N;
    for (int64_t p = beg + gl; p < end; p += G) narrow_fma<F>(acc, val[p], X + (xoff + col[p]) * F);
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) {
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] += __shfl_xor(acc[f], o);
  }
  if (live && !is_long && gl == 0) {
#pragma unroll
    for (int f = 0; f < F; ++f) Y[r * F + f] = acc[f];
  }
  if constexpr (G < kWave) {
    const int lane = threadIdx.x & 63;
    for (uint64_t m = __ballot(is_long && gl == 0); m; m &= m - 1) {
      const int src = __builtin_ctzll(m);
      const int64_t r2 = readlane64(r, src), b2 = readlane64(beg, src), e2 = readlane64(end, src);
      narrow_wave_row<F>(acc, col, val, X, (r2 / N) * (int64_t)N, b2, e2, lane);
      if (lane == 0) {
#pragma unroll
        for (int f = 0; f < F; ++f) Y[r2 * F + f] = acc[f];
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// generic kernel: any F.  LPR lanes across F (scalar), S streams over non-zeros.
// ---------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void spmm_generic_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ X, float* __restrict__ Y,
    int64_t n_rows, int32_t N, int32_t F) {
  constexpr int S = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR;
  const int fl = lane % LPR;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;  // whole wave exits together
  const int64_t beg = rowptr[r];
  const int64_t end = rowptr[r + 1];
  const int64_t xoff = (r / N) * (int64_t)N;
  for (int f0 = 0; f0 < F; f0 += LPR) {
    const int f = f0 + fl;
    float acc = 0.f;
    for (int64_t p = beg + sub; p < end; p += S) {
      const float v = val[p];
      const int64_t src = (xoff + col[p]) * F;
      if (f < F) acc = fmaf(v, X[src + f], acc);
    }
#pragma unroll
    for (int o = LPR; o < kWave; o <<= 1) acc += __shfl_xor(acc, o);
    if (sub == 0 && f < F) Y[r * F + f] = acc;
  }
}

template <int F>
static int launch_small(const int64_t* rowptr, const int32_t* col, const float* val,
                        const float* X, float* Y, int64_t n_rows, int32_t N, int G,
                        hipStream_t st) {
  const int64_t threads = n_rows * G;
  const unsigned grid = (unsigned)((threads + 255) / 256);
#define TMGCN_SMALL_CASE(GG)                                                              \
  case GG:                                                                                \
    hipLaunchKernelGGL((spmm_small_kernel<F, GG>), dim3(grid), dim3(256), 0, st, rowptr,  \
                       col, val, X, Y, n_rows, N);                                        \
    break;
  switch (G) {
    TMGCN_SMALL_CASE(1)
    TMGCN_SMALL_CASE(2)
    TMGCN_SMALL_CASE(4)
    TMGCN_SMALL_CASE(8)
    TMGCN_SMALL_CASE(16)
    TMGCN_SMALL_CASE(32)
    default:
      hipLaunchKernelGGL((spmm_small_kernel<F, 64>), dim3(grid), dim3(256), 0, st, rowptr, col,
                         val, X, Y, n_rows, N);
  }
#undef TMGCN_SMALL_CASE
  return check_launch("spmm_small");
}

}  // namespace tmgcn

using namespace tmgcn;

// avg_nnz_per_row hint: < 0 means "unknown" (G defaults to 8).  Exposed through a
// second entry point so the public signature stays the reference-shaped one.
extern "C" int64_t tmgcn_spmm_giant_workspace_bytes(int32_t n_giant_chunks, int32_t F) {
  return n_giant_chunks > 0 && F > 0 ? (int64_t)n_giant_chunks * F * (int64_t)sizeof(float) : 0;
}

extern "C" int tmgcn_spmm_csr_batched_f32_plan(const int64_t* rowptr, const int32_t* col,
                                                const float* val, const float* X, float* Y,
                                                int64_t n_rows, int32_t N, int32_t F,
                                                float avg_nnz_per_row, const int64_t* giant_rows,
                                                const int32_t* giant_chunks, int32_t n_giant, int32_t n_giant_chunks,
                                                float* giant_ws, int64_t giant_ws_bytes, void* stream) {
  TMGCN_REQUIRE(n_rows >= 0 && N > 0 && F > 0, "spmm: bad shape n_rows=%lld N=%d F=%d",
                (long long)n_rows, N, F);
  if (n_rows == 0) return TMGCN_OK;
  TMGCN_REQUIRE(rowptr && X && Y, "spmm: null pointer");
  TMGCN_REQUIRE(n_rows % N == 0, "spmm: n_rows=%lld is not a multiple of N=%d",
                (long long)n_rows, N);
  hipStream_t st = (hipStream_t)stream;

  // lanes-over-nnz group size for the small-F path
  int G = 8;
  if (avg_nnz_per_row >= 0.f) {
    G = 1;
    while (G < 64 && (float)(2 * G) <= avg_nnz_per_row) G <<= 1;
  }

  switch (F) {
    case 1: return launch_small<1>(rowptr, col, val, X, Y, n_rows, N, G, st);
    case 2: return launch_small<2>(rowptr, col, val, X, Y, n_rows, N, G, st);
    case 3: return launch_small<3>(rowptr, col, val, X, Y, n_rows, N, G, st);
    case 4: return launch_small<4>(rowptr, col, val, X, Y, n_rows, N, G, st);
    case 6: return launch_small<6>(rowptr, col, val, X, Y, n_rows, N, G, st);
    case 8: return launch_small<8>(rowptr, col, val, X, Y, n_rows, N, G, st);
    default: break;
  }

  if (F % 4 == 0 && F >= 16 && F <= 4096 &&
      (reinterpret_cast<uintptr_t>(X) % 16 == 0) && (reinterpret_cast<uintptr_t>(Y) % 16 == 0)) {
    const int F4 = F / 4;
    int lpr = 4;
    while (lpr < F4 && lpr < 64) lpr <<= 1;                 // F > 256: 64 lanes, column chunks of 256 floats
    const int64_t n_tiles = make_tile_map(n_rows, N).n_tiles;
    TMGCN_REQUIRE(n_tiles < (int64_t)0x7fffffff, "spmm: too many row tiles");
    GiantPlan giant{nullptr, nullptr, nullptr, 0};
    if (n_giant > 0) {
      const int rc = launch_giant_partial("spmm (giant rows)", rowptr, col, val, X, N, F, giant_rows, giant_chunks, n_giant,
                                          n_giant_chunks, giant_ws, giant_ws_bytes, st);
      if (rc != TMGCN_OK) return rc;
      giant = GiantPlan{giant_rows, giant_chunks, reinterpret_cast<const float4*>(giant_ws), n_giant};
    }
    unsigned int* counter = acquire_tile_counters(st, 2);       // [0] tiles, [1] heavy-tile scan windows
    TMGCN_REQUIRE(counter, "spmm: no tile counter: %s", pool_error());
    const float4* X4 = reinterpret_cast<const float4*>(X);
    float4* Y4 = reinterpret_cast<float4*>(Y);
#define TMGCN_VEC_CASE(L, UU)                                                                \
  case L: {                                                                                  \
    int64_t gx = 2 * (int64_t)persistent_grid(spmm_vec4_kernel<L, UU, TMGCN_SPMM_US * UU>, 256);         \
    if (gx > n_tiles) gx = n_tiles;                                                          \
    hipLaunchKernelGGL((spmm_vec4_kernel<L, UU, TMGCN_SPMM_US * UU>), dim3((unsigned)gx), dim3(256), 0, st, \
                       rowptr, col, val, X4, Y4, n_rows, N, F4, counter, giant);             \
    break;                                                                                   \
  }
    switch (lpr) {
      TMGCN_VEC_CASE(4, 2)
      TMGCN_VEC_CASE(8, 2)
      TMGCN_VEC_CASE(16, TMGCN_SPMM_U)
      TMGCN_VEC_CASE(32, TMGCN_SPMM_U)
      TMGCN_VEC_CASE(64, TMGCN_SPMM_U)
    }
#undef TMGCN_VEC_CASE
    return check_launch("spmm_vec4");
  }

  {
    int lpr = 1;
    while (lpr < F && lpr < 64) lpr <<= 1;
    const unsigned grid = (unsigned)((n_rows + 3) / 4);
#define TMGCN_GEN_CASE(L)                                                                    \
  case L:                                                                                    \
    hipLaunchKernelGGL((spmm_generic_kernel<L>), dim3(grid), dim3(256), 0, st, rowptr, col,  \
                       val, X, Y, n_rows, N, F);                                             \
    break;
    switch (lpr) {
      TMGCN_GEN_CASE(1)
      TMGCN_GEN_CASE(2)
      TMGCN_GEN_CASE(4)
      TMGCN_GEN_CASE(8)
      TMGCN_GEN_CASE(16)
      TMGCN_GEN_CASE(32)
      TMGCN_GEN_CASE(64)
    }
#undef TMGCN_GEN_CASE
    return check_launch("spmm_generic");
  }
}

extern "C" int tmgcn_spmm_csr_batched_f32_hint(const int64_t* rowptr, const int32_t* col,
                                                const float* val, const float* X, float* Y,
                                                int64_t n_rows, int32_t N, int32_t F,
                                                float avg_nnz_per_row, void* stream) {
  return tmgcn_spmm_csr_batched_f32_plan(rowptr, col, val, X, Y, n_rows, N, F, avg_nnz_per_row, nullptr, nullptr, 0, 0,
                                         nullptr, 0, stream);
}

extern "C" int tmgcn_spmm_csr_batched_f32(const int64_t* rowptr, const int32_t* col,
                                           const float* val, const float* X, float* Y,
                                           int64_t n_rows, int32_t N, int32_t F, void* stream) {
  return tmgcn_spmm_csr_batched_f32_hint(rowptr, col, val, X, Y, n_rows, N, F, -1.f, stream);
}

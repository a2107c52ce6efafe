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
// `stride4` = row pitch of Xs in float4 (default: F4, a dense [N][F] matrix); a wider matrix is gathered as column chunks.
template <int LPR, int U>
__device__ __forceinline__ float4 gather_row(const int32_t* __restrict__ col,
                                             const float* __restrict__ val,
                                             const float4* __restrict__ Xs, int64_t beg,
                                             int64_t end, int F4, int lane, int stride4 = 0) {
  if (stride4 == 0) stride4 = F4;
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
        if (idx < n && f_ok) x[u] = Xs[(int64_t)cc * stride4 + fl];
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

// ------------------------------------------------------------------------------------------------------------
// Skewed row lengths (the reference's real operand is the M-product of symmetrised real graphs, read_data.py:116-127,
// 204-223: a few rows hold most of the entries).  Two measures, shared by the plain and the GEMM-fused kernel, neither
// needs a plan or an atomic and both keep every row sum in a fixed order:
//   * a row longer than kLongRow entries is gathered by ALL FOUR waves of the block — consecutive quarters of its
//     entry range (a multiple of 64 each), the four partial sums added in wave order through LDS — instead of
//     keeping one wave busy while the tile's other rows and the block's MFMA phase wait for it;
//   * tiles with more than max(kHeavyMin, 8x the launch's mean) entries are taken FIRST: before a block starts
//     drawing tiles from the launch's counter it draws windows of 64 consecutive tiles from a second counter, reads
//     their extents (one coalesced load pair) and processes the heavy ones it finds; the counter-driven loop then skips
//     them (same predicate, recomputed from the row extents it loads anyway).  A 100 000-entry hub — milliseconds even on four waves —
//     therefore runs at the start of the launch, under everything else, never as its tail.
// ------------------------------------------------------------------------------------------------------------
#ifndef TMGCN_LONG_ROW
#define TMGCN_LONG_ROW 256
#endif
#ifndef TMGCN_HEAVY_FIRST
#define TMGCN_HEAVY_FIRST 1
#endif
constexpr int kTileRows = 64;                 // rows per tile of both kernels
constexpr int kLongRow = TMGCN_LONG_ROW;
constexpr int64_t kHeavyMin = 8192;

__device__ __forceinline__ int64_t readlane64(int64_t v, int l) {   // l wave-uniform
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffff), l);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((int64_t)hi << 32) | lo;
}

// tile -> rows.  Tiles restart at every UNIT of rows — a slice (N rows) in every launch the layers make — in the plain and in
// the fused kernel alike: a tile never straddles two slices, and a row falls into the same tile whichever kernel, weight
// layout (shared / per slice) or slice window (the pipelined one-slice launches of dist.py) it is summed under.  Which path
// sums a row (row per wave, four waves, entry-major) is a function of its tile, so this is what keeps all of them bit-equal.
struct TileMap {
  int64_t n_rows, unit_rows, tiles_per_unit, n_tiles;
};
__host__ __device__ __forceinline__ TileMap make_tile_map(int64_t n_rows, int64_t unit_rows) {
  const int64_t per = (unit_rows + 63) / 64;
  return TileMap{n_rows, unit_rows, per, ((n_rows + unit_rows - 1) / unit_rows) * per};
}
// first row of the tile and the end of its unit (the tile holds rows [row0, min(row0 + 64, row_end)))
__device__ __forceinline__ void tile_extent(const TileMap& m, int64_t tile, int64_t& unit, int64_t& row0, int64_t& row_end) {
  unit = tile / m.tiles_per_unit;
  row0 = unit * m.unit_rows + (tile - unit * m.tiles_per_unit) * kTileRows;
  row_end = (unit + 1) * m.unit_rows;
  if (row_end > m.n_rows) row_end = m.n_rows;
}

// Pass 1: the heavy tiles, found 64 extents at a time.  Blocks draw WINDOWS of 64 consecutive tiles from a second device
// counter (counter[1]; counter[0] feeds the main loop) until the windows run out, and process the heavy tiles of the windows
// they drew: whichever blocks are resident first share ALL the heavy tiles — a block that becomes resident late (CU-masked
// streams, RCCL kernels holding CUs) finds nothing left to scan instead of owning a share that would run as the launch's
// tail.  All state is block-uniform (every wave computes the same ballot from the same loads).
struct HeavyScan {
  int64_t thr, win;
  uint64_t pending;
  bool scanning;
  __device__ __forceinline__ void init(const int64_t* __restrict__ rowptr, const TileMap& m) {
    const int64_t mean = (rowptr[m.n_rows] - rowptr[0]) / m.n_tiles;
    thr = 8 * mean > kHeavyMin ? 8 * mean : kHeavyMin;
    win = 0;
    pending = 0;
    scanning = TMGCN_HEAVY_FIRST != 0;
  }
  // next heavy tile of this block, or -1 when the scan is over (then `scanning` is false); s_slot: one LDS word of the block
  __device__ __forceinline__ int64_t next(const int64_t* __restrict__ rowptr, const TileMap& m, unsigned int* scan_counter,
                                          unsigned int* s_slot, int lane) {
    while (pending == 0) {
      if (threadIdx.x == 0) *s_slot = atomicAdd(scan_counter, 1u);
      __syncthreads();
      win = (int64_t)*s_slot * kWave;
      __syncthreads();                                 // everybody has read the slot before it is written again
      if (win >= m.n_tiles) {
        scanning = false;
        return -1;
      }
      const int64_t t = win + lane;
      int64_t ent = 0;
      if (t < m.n_tiles) {
        int64_t b, r0, r1;
        tile_extent(m, t, b, r0, r1);
        if (r0 + kTileRows < r1) r1 = r0 + kTileRows;
        ent = rowptr[r1] - rowptr[r0];
      }
      pending = __ballot(ent > thr);
    }
    const int l = __builtin_ctzll(pending);
    pending &= pending - 1;
    return win + l;
  }
};

// Extents of the tile's rows, one coalesced load pair per wave: lane l holds row row0 + l (empty past row_end).
struct TileRows {
  int64_t beg, end;
  uint64_t long_mask;   // rows with more than kLongRow entries
  int64_t entries;      // of the whole tile
  __device__ __forceinline__ void load(const int64_t* __restrict__ rowptr, int64_t row0, int64_t row_end, int lane) {
    const int64_t r = row0 + lane;
    beg = 0;
    end = 0;
    if (r < row_end) {
      beg = rowptr[r];
      end = rowptr[r + 1];
    }
    long_mask = __ballot(end - beg > kLongRow);
    const int last = (int)((row_end - row0 < kTileRows ? row_end - row0 : kTileRows) - 1);
    entries = readlane64(end, last) - readlane64(beg, 0);
  }
};

// A long row on four waves: wave w gathers quarter w, the partial sums meet in `part` ([4][LPR] float4 of LDS) and
// lanes < LPR of EVERY wave return ((p0 + p1) + p2) + p3.  Called by all 256 threads; two block barriers.
template <int LPR, int U>
__device__ __forceinline__ float4 gather_long_row(const int32_t* __restrict__ col, const float* __restrict__ val,
                                                  const float4* __restrict__ Xs, int64_t beg, int64_t end, int F4, int lane,
                                                  int wave, float4* part, int stride4 = 0) {
  const int64_t q = (((end - beg + 3) >> 2) + (kWave - 1)) & ~(int64_t)(kWave - 1);
  int64_t b = beg + wave * q, e = b + q;
  if (b > end) b = end;
  if (e > end) e = end;
  const float4 p = gather_row<LPR, U>(col, val, Xs, b, e, F4, lane, stride4);
  if (lane < LPR) part[wave * LPR + lane] = p;
  __syncthreads();
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < LPR) {
    s = part[lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = part[w * LPR + lane];
      s.x += t.x;
      s.y += t.y;
      s.z += t.z;
      s.w += t.w;
    }
  }
  __syncthreads();
  return s;
}

// ------------------------------------------------------------------------------------------------------------
// Short tiles (round 6).  The reference's real operand — the M-product of its symmetrised, windowed chess slices,
// read_data.py:116-127, 204-223 — holds 4 entries per row on average and the self loop alone in two rows of three.  One wave
// per row then spends two dependent round trips (col / val, then ONE gather on half of its lanes) per row, sixteen rows one
// after the other per wave and tile: the launch is a latency chain (measured: fused forward 23.6 ms where its compulsory bytes
// are 6.3 ms of HBM time and its products 6.7 ms of MFMA time).  A tile of at most kShortTile entries without a long row is
// therefore walked ENTRY-major: each of the block's 4·S lane groups (S = 64/LPR per wave) takes consecutive whole rows — an
// equal share of the tile's ENTRIES, up to one row — i.e. a CONTIGUOUS run of entries, fetched LPR at a time with one
// coalesced load per array; every lane finds the row of the entry it loaded by bisecting the tile's row ends (which every
// wave holds, one per lane); the group then walks its entries in order,
// U gathers in flight whatever rows they belong to, and hands a row's sum to `flush` when the row changes.  A row is summed
// by ONE group in entry order (an fmaf chain: bitwise the serial sum); no atomics, no LDS, reproducible.  The predicate is a
// function of the tile's row pointers alone, so the plain and the fused kernel, and every rerun, take the same path.
// ------------------------------------------------------------------------------------------------------------
#ifndef TMGCN_SHORT_TILE
#define TMGCN_SHORT_TILE 512
#endif
constexpr int kShortTile = TMGCN_SHORT_TILE;      // entries: a mean of at most 8 per row

// flush(tile_row, sum, fl): called by the LPR lanes of ONE group (lane fl of the group holds float4 fl of the sum), exactly
// once for every row r < n_tile_rows of the tile over the block's four waves.  Xs = X of the tile's slice (a short tile lies
// in ONE slice).
// The shuffles that hand out an entry's value and row run AFTER the batch's gathers are issued: only the U float4 in
// flight and the lane's own (col, val, row) are live across the loads.
template <int LPR, int U, int NW = 4, class Flush>        // NW: the waves that share the tile (4; the bf16-product kernel: 8)
__device__ __forceinline__ void gather_short_tile(const int32_t* __restrict__ col, const float* __restrict__ val,
                                                  const float4* __restrict__ Xs, const TileRows& rows, int n_tile_rows,
                                                  int F4, int lane, int wave, int stride4, Flush&& flush) {
  constexpr int S = kWave / LPR;                   // lane groups per wave
  constexpr int NG = NW * S;                       // ... per block
  const int sub = lane / LPR;
  const int fl = lane % LPR;
  const bool f_ok = fl < F4;
  // row extents relative to the tile's first entry (<= kShortTile: int32); rows behind the tile's last are empty at its end
  const int64_t tile_beg = readlane64(rows.beg, 0);
  const int tile_ent = (int)rows.entries;
  const int rb = lane < n_tile_rows ? (int)(rows.beg - tile_beg) : tile_ent;
  const int re = lane < n_tile_rows ? (int)(rows.end - tile_beg) : tile_ent;
  // The rows are dealt to the block's NG lane groups by ENTRIES: group G takes the rows that START in the G-th NG-th of the
  // tile's entry range — consecutive rows, whole rows, at most one row's length off an equal share (a fixed sixteen rows per
  // wave left the wave with the tile's one 60-entry row 2.3 us behind the others at the barrier).  The first row of group G =
  // the number of rows that start before its share does: one ballot per boundary.
  int gb = tile_ent, ge = tile_ent;
  {
    int r0 = 0, r1 = 0;
#pragma unroll
    for (int s = 0; s <= S; ++s) {
      const int t = (int)(((int64_t)(wave * S + s) * tile_ent + NG - 1) / NG);        // uniform
      const int cnt = __popcll(__ballot(rb < t));                                     // rows that start before t
      if (s == sub) r0 = cnt;
      if (s == sub + 1) r1 = cnt;
    }
    const int b_at_r0 = __shfl(rb, r0 & 63), b_at_r1 = __shfl(rb, r1 & 63);        // every lane takes part
    gb = r0 < kWave ? b_at_r0 : tile_ent;
    ge = r1 < kWave ? b_at_r1 : tile_ent;
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int cur = -1;
  for (int b0 = gb; __any(b0 < ge); b0 += LPR) {
    const int e = b0 + fl;                         // the entry this lane fetches for its group
    int c = 0, rid = 0;
    float v = 0.f;
    if (e < ge) {
      c = col[tile_beg + e];
      v = val[tile_beg + e];
    }
    // the row of entry e = the number of rows that end at or before e (the row ends ascend): a branch-free bisection over
    // the 64 ends the lanes hold
#pragma unroll
    for (int step = kWave / 2; step > 0; step >>= 1) rid += (__shfl(re, rid + step - 1) <= e) ? step : 0;
    const int nb = ge - b0 < LPR ? ge - b0 : LPR;  // entries of this batch (<= 0: the group is done)
    for (int p = 0; __any(p < nb); p += U) {
      float4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cc = __shfl(c, sub * LPR + ((p + u) & (LPR - 1)));
        x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p + u < nb && f_ok) x[u] = Xs[(int64_t)cc * stride4 + fl];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = sub * LPR + ((p + u) & (LPR - 1));
        const float vv = __shfl(v, src);
        const int rr = __shfl(rid, src);
        if (p + u < nb) {
          if (rr != cur) {
            if (cur >= 0) flush(cur, acc, fl);
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            cur = rr;
          }
          acc.x = fmaf(vv, x[u].x, acc.x);
          acc.y = fmaf(vv, x[u].y, acc.y);
          acc.z = fmaf(vv, x[u].z, acc.z);
          acc.w = fmaf(vv, x[u].w, acc.w);
        }
      }
    }
  }
  if (cur >= 0) flush(cur, acc, fl);
  // rows without entries are never met by the walk: their sum is zero
  const uint64_t empty = __ballot(lane < n_tile_rows && rows.end == rows.beg) & ((~0ull >> (kWave - kWave / NW)) << (kWave / NW * wave));
  for (uint64_t m = empty; m; m &= m - 1)
    if (sub == 0) flush(__builtin_ctzll(m), make_float4(0.f, 0.f, 0.f, 0.f), fl);
}

// Does the tile take the entry-major walk?  Block-uniform (every wave holds the same TileRows).  `one_slice`: all its rows
// gather from the same slice of X (always, when tiles restart at every slice: TileMap).
__device__ __forceinline__ bool short_tile(const TileRows& rows, bool one_slice) {
  return kShortTile > 0 && rows.long_mask == 0 && rows.entries <= kShortTile && one_slice;
}

// ------------------------------------------------------------------------------------------------------------
// Narrow kernels (F <= 8: spmm_small, spmm_gemm_small): G lanes share a row and stride over its entries.  A row that
// would take a group more than kNarrowLong trips is left to the WHOLE WAVE afterwards: the groups flag their long rows
// (ballot), and for each of them all 64 lanes stride over the entries, four (col, val) pairs and four gathers in flight per
// lane, combined by the full butterfly — 8 to 64 times the lanes and a quarter of the dependent round trips for the rows
// that would otherwise be the launch's tail (a 10 000-entry hub on 8 lanes: 1 250 trips).  Fixed order: reproducible.
// ------------------------------------------------------------------------------------------------------------
constexpr int kNarrowLong = 32;

template <int F>
__device__ __forceinline__ void narrow_fma(float (&acc)[F], float v, const float* __restrict__ x) {
  if constexpr (F % 4 == 0) {
#pragma unroll
    for (int q = 0; q < F / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(x + 4 * q);
      acc[4 * q + 0] = fmaf(v, t.x, acc[4 * q + 0]);
      acc[4 * q + 1] = fmaf(v, t.y, acc[4 * q + 1]);
      acc[4 * q + 2] = fmaf(v, t.z, acc[4 * q + 2]);
      acc[4 * q + 3] = fmaf(v, t.w, acc[4 * q + 3]);
    }
  } else if constexpr (F % 2 == 0) {
#pragma unroll
    for (int q = 0; q < F / 2; ++q) {
      const float2 t = *reinterpret_cast<const float2*>(x + 2 * q);
      acc[2 * q + 0] = fmaf(v, t.x, acc[2 * q + 0]);
      acc[2 * q + 1] = fmaf(v, t.y, acc[2 * q + 1]);
    }
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = fmaf(v, x[f], acc[f]);
  }
}

// Σ over [beg, end) of val[p]·X[xoff + col[p]] by the 64 lanes of the calling wave (beg, end wave-uniform, end > beg); on
// return every lane holds the sum.
template <int F>
__device__ __forceinline__ void narrow_wave_row(float (&acc)[F], const int32_t* __restrict__ col, const float* __restrict__ val,
                                                const float* __restrict__ X, int64_t xoff, int64_t beg, int64_t end, int lane) {
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  constexpr int NB = 4;
  for (int64_t p = beg + lane; p < end; p += NB * kWave) {
    int c[NB];
    float v[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {                      // unconditional loads on clamped positions, masked by a zero value
      const int64_t q = p + u * kWave;
      const int64_t qc = q < end ? q : end - 1;
      c[u] = col[qc];
      v[u] = q < end ? val[qc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) narrow_fma<F>(acc, v[u], X + (xoff + c[u]) * F);
  }
#pragma unroll
  for (int o = kWave >> 1; o > 0; o >>= 1)
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] += __shfl_xor(acc[f], o);
}

// ------------------------------------------------------------------------------------------------------------
// Giant rows.  Four waves of one block gather 41 M entries/s (tools/hub_tail_probe.py): a row of 10^6 entries is 23 ms
// on its own, whatever else the launch holds.  With a PLAN from the caller (the rows of more than kGiantRow entries and a
// chunk count for each: csr.BatchedCSR.giant_plan) such rows are cut into chunks of kGiantChunk entries that a small
// launch in front of the main kernel sums one block per chunk (spmm_giant_partial_kernel, spmm.hip: the same four-wave
// gather, partial sums to a workspace), and the main kernels ADD UP a giant row's partial sums, in chunk order, instead of
// gathering it.  Stream order is the only synchronisation; sums stay in a fixed order.  Without a plan (NULL) a giant
// row takes the four-wave path like any other long row.
// ------------------------------------------------------------------------------------------------------------
constexpr int kGiantRow = TMGCN_GIANT_ROW;        // include/tmgcn.h
constexpr int kGiantChunk = TMGCN_GIANT_CHUNK;

struct GiantPlan {
  const int64_t* rows;        // [n] ascending global row indices (k*N + i) of the rows with more than kGiantRow entries
  const int32_t* chunk_ptr;   // [n + 1] first chunk of each of them
  const float4* partial;      // [n_chunks][F4] partial sums
  int32_t n;
};

// Position of row r in the plan, or -1 when the plan does not list it (a plan built for another CSR or with another
// threshold: the caller then gathers the row like any other long row instead of adding up somebody else's partial sums).
// Wave-uniform; called by every wave of the block (a few L2 hits, once per giant row).
__device__ __forceinline__ int giant_find(const GiantPlan& g, int64_t r) {
  int lo = 0, hi = g.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (g.rows[mid] < r) lo = mid + 1;
    else hi = mid;
  }
  return (g.n > 0 && g.rows[lo] == r) ? lo : -1;
}

// Σ of the partial sums of the plan's giant row `gi` (giant_find); lanes < w4 of the calling wave return their float4 of
// columns [4·(c0 + lane), …); wave-uniform control flow.
__device__ __forceinline__ float4 giant_row_sum(const GiantPlan& g, int gi, int F4, int lane, int c0, int w4) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < w4) {
    const int c_end = g.chunk_ptr[gi + 1];
    for (int c = g.chunk_ptr[gi]; c < c_end; ++c) {
      const float4 t = g.partial[(int64_t)c * F4 + c0 + lane];
      s.x += t.x;
      s.y += t.y;
      s.z += t.z;
      s.w += t.w;
    }
  }
  return s;
}

}  // namespace tmgcn

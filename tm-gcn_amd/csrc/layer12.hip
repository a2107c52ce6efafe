// Layers 1 + 2 of the narrow 2-layer models in one forward and one backward kernel   (gfx950 / CDNA4)
//
//     Y = act1(H · W1)                  ehf:330-335 (EmbeddingGCN2, H = the cached AtXt), ehf:486 (EmbeddingKWGCN, H = AX)
//     Z = act2((Â ⋆ Y) · W2)            ehf:348-349 (the as-run default branch: the TRAINING adjacency), ehf:487
// for the reference's own widths (F0 = 2 -> 6 -> 6): H [R][2] is a constant of the model (cached at
// construction, ehf:293 / 464), so Y never has to exist.  At these sizes (R = T·N = 570 k rows, 3 non-zeros per
// row) the separate kernels are bound by the [R][6] tensors they pass to each other, not by arithmetic:
//   forward   the SpMM gathers 8-byte H rows instead of 24-byte Y rows and applies W1 and the non-linearity to
//             each gathered row on the fly (12 fmas + 6 activations per non-zero) — gemm_small's fmaf chain for
//             layer 1; a row's non-zeros are summed NB consecutive ones per lane and folded over the lanes of the row,
//             spmm_gemm_small's strided over its lanes: two fp32 summation orders, 3e-7 apart at 3 per row, 1e-6 at 27;
//   backward  dY = (Âᵀ ⋆ dZ) · W2ᵀ per row, P = H·W1 recomputed (12 fmas), dP = dY ⊙ act1'(P), and
//             dW1 = Σ_r H[r]ᵀ·dP[r] accumulated on the spot (fp64, dealt over the lanes of a row, slabs reduced by the
//             last block): no dY, no pre-activation and no dP tensor, no separate dW1 launch.
// Small, dense slices (N·width·4 B <= 64 KB and >= 8 non-zeros per row: the AMLSim shape, N = 1 000, 27 per row) take the
// STAGED variants: a block belongs to one slice, first forms what its rows will gather — act1(H·W1) of every node of the
// slice (forward), dZ ⊙ act2'(pre2) of every node (backward) — in LDS, once per node instead of once per non-zero,
// and then gathers from LDS: 24-byte rows at random addresses cost an L1 tag look-up per lane and instruction through
// the vector memory path (measured 6.5 clocks per non-zero per CU) and a fraction of that from LDS.
// Three walks of the rows, chosen per call from the shape (tmgcn_layer12_fwd_f32 / _bwd_f32):
//   entry-major   slices of >= 256 nodes that the staged variants do not take — the forward at every density, the backward
//                 for sparse rows (< 4 non-zeros per row on average) and whenever the caller brings a partition of the rows
//                 (skewed adjacencies): a block walks the contiguous entry range of a row block — 256 rows, or the caller's
//                 (first row, rows) pair — tile by tile and sums each row from LDS (l12_fwd_em_kernel: one block per row
//                 block; l12_bwd_em_kernel: resident blocks that draw row blocks) — three dependent round trips per row
//                 block whatever the row lengths; the Bitcoin-OTC shape and real, skewed data;
//   staged        small dense slices (above);
//   lanes per row everything else: G = 1 … 16 lanes per row, entries strided over them (l12_fwd_kernel, l12_bwd_kernel).
// dW2 = (Â⋆Y)ᵀ·(dZ ⊙ act2'(pre2)): the ENTRY-MAJOR backward forms it in the same launch (round 5: four lanes per row load the
// row block's own rows of the two [R][6] tensors while the row pointers are in flight, the wave folds the 3 x 3 quadrant
// products, the sums ride in the row block's slab; tmgcn_layer12_bwd_forms_dw2 tells a caller when) — the first attempt,
// 48 fp64 accumulators per LANE, had cost 72 us against 37 + 13 and was not kept.  The staged and the lanes-per-row
// backward leave dW2 to the narrow dW kernel (gemm.hip) on the Â⋆Y the forward stores for it.
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
  float* dW2;              // backward, entry-major kernel only, optional: [F][NT] = AXᵀ·(dZ ⊙ act2'(pre2)) (AX then holds the forward's Â⋆Y)
  float* part;             // backward: [blocks][KI·F] slabs (entry-major: one per row block, + F·NT with dW2)
  int32_t* sync;
  int64_t n_rows;
  int32_t N;
  int32_t act1, act2;
  int32_t chunks, chunk_rows;   // staged variants: blocks per slice and rows per block
  const int64_t* blk;           // entry-major variants, optional: (first row, rows <= 256) of every row block, n_blk pairs
  int32_t n_blk;
};

__device__ __forceinline__ int64_t l12_readlane64(int64_t v, int l) {   // l wave-uniform
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffff), l);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((int64_t)hi << 32) | lo;
}

template <int N, typename T>
__device__ __forceinline__ T pick_at(const T (&v)[N], int i) {
  T r = v[0];
#pragma unroll
  for (int q = 1; q < N; ++q) r = (i == q) ? v[q] : r;
  return r;
}

// NB consecutive non-zeros starting at `base`, as unaligned 16-byte loads (four column indices / four values each) while they
// lie inside the arrays, element loads clamped to the last entry otherwise (the tail of the last rows only).  Every load
// is unconditional and every column index returned is a valid one: the caller skips the positions past its row's end.
// Development build only (-DTMGCN_L12_TRACE, tools/l12_trace.py): thread 0 of every block leaves 100 MHz wall-clock stamps of
// the entry-major backward's phases in a device array, read back through tmgcn_debug_l12_trace.  Not part of the library.
#ifdef TMGCN_L12_TRACE
__device__ unsigned long long l12_trace_words[8192 * 16];
__device__ unsigned long long l12_trace_fwd_words[8192 * 16];
#define L12_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192 && (i) < 16) l12_trace_words[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#define L12_STAMP_FWD(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192 && (i) < 16) l12_trace_fwd_words[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define L12_STAMP(i) do { } while (0)
#define L12_STAMP_FWD(i) do { } while (0)
#endif

struct __attribute__((packed, aligned(4))) Int4u { int32_t v[4]; };
struct __attribute__((packed, aligned(4))) Float4u { float v[4]; };

template <int NB>
__device__ __forceinline__ void load_entries(const int32_t* col, const float* val, int64_t base, int64_t nnz, int (&c)[NB], float (&v)[NB]) {
  static_assert(NB % 4 == 0, "four entries per load");
  if (base + NB <= nnz) {
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
      const Int4u ci = *reinterpret_cast<const Int4u*>(col + base + 4 * q);
      const Float4u vi = *reinterpret_cast<const Float4u*>(val + base + 4 * q);
#pragma unroll
      for (int u = 0; u < 4; ++u) c[4 * q + u] = ci.v[u], v[4 * q + u] = vi.v[u];
    }
  } else {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int64_t q = base + u < nnz ? base + u : nnz - 1;
      c[u] = col[q];
      v[u] = val[q];
    }
  }
}

// The same for entries STRIDED over the G lanes of a row (lane gl takes t0 + gl + u·G): element loads, clamped to the row's
// last entry.  Neighbouring lanes then hold neighbouring entries, whose columns — sorted inside a row — are close: the
// gathers of a group fall into few cache lines.  Used by the unstaged kernels when a row has several lanes (the staged ones
// gather from LDS, one-lane rows have no neighbours): backward at 16 / 8 / 40 non-zeros per row of 8 000 / 7 301 / 20 000 nodes
// 97 / 54 / 268 us against 108 / 57 / 302 with consecutive entries per lane (rocprofv3 kernel stats, tools/l12_shape_profile.py).
template <int NB, int G>
__device__ __forceinline__ void load_entries_strided(const int32_t* col, const float* val, int64_t first, int64_t end, int (&c)[NB], float (&v)[NB]) {
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int64_t q = first + u * G < end ? first + u * G : end - 1;
    c[u] = col[q];
    v[u] = val[q];
  }
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

extern __shared__ float l12_stage[];   // staged variants: [N][F] (forward) / [N][NT] (backward) of the block's slice

template <int KI, int F, int NT, int G, bool STAGED>
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
  const int64_t nnz = a.rowptr[a.n_rows];
  static_assert(KI == 2, "H rows are float2");
  const int gl = threadIdx.x & (G - 1);
  int64_t r, r_end, xoff;
  int64_t r_base = 0;
  if constexpr (STAGED) {
    const int slice = blockIdx.x / a.chunks, chunk = blockIdx.x - slice * a.chunks;
    xoff = (int64_t)slice * a.N;
    for (int c = threadIdx.x; c < a.N; c += 256) {               // layer 1 of every node of the slice, once
      const float2 hv = *reinterpret_cast<const float2*>(a.H + (xoff + c) * KI);
      const float h[KI] = {hv.x, hv.y};
      float y[F];
      layer1_row<KI, F>(h, W1, act1, y);
#pragma unroll
      for (int f = 0; f < F; f += 2) *reinterpret_cast<float2*>(l12_stage + c * F + f) = make_float2(y[f], y[f + 1]);
    }
    const int first = chunk * a.chunk_rows, last = first + a.chunk_rows < a.N ? first + a.chunk_rows : a.N;
    int64_t* rps = reinterpret_cast<int64_t*>(l12_stage + a.N * F);      // the block's row pointers
    for (int i = threadIdx.x; i <= last - first; i += 256) rps[i] = a.rowptr[xoff + first + i];
    r_base = xoff + first;
    __syncthreads();
    r = xoff + first + threadIdx.x / G;
    r_end = xoff + last;
  } else {
    r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    r_end = r < a.n_rows ? r + 1 : r;                            // one row per group
    xoff = (r / a.N) * (int64_t)a.N;
  }
  for (; r < r_end; r += 256 / G) {
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
    int64_t beg, end;
    if constexpr (STAGED) {
      const int64_t* rps = reinterpret_cast<const int64_t*>(l12_stage + a.N * F);
      beg = rps[r - r_base], end = rps[r - r_base + 1];
    } else {
      beg = a.rowptr[r], end = a.rowptr[r + 1];
    }
    // A trip takes NB·G non-zeros of the row, NB CONSECUTIVE ones per lane: their (col, val) pairs are requested together
    // (16-byte loads) and then their H rows together — two dependent round trips per trip instead of two per non-zero,
    // and a quarter of the load instructions and address arithmetic of element loads.  All loads are unconditional (a
    // load under a per-lane condition becomes a branch with a full wait); positions past the row's end are skipped in
    // the arithmetic.
    constexpr int NB = STAGED ? 8 : 4;
    for (int64_t t0 = beg; t0 < end; t0 += NB * G) {
      constexpr bool STRIDED = !STAGED && G > 1;
      constexpr int STEP = STRIDED ? G : 1;                   // entry of slot u: base + u·STEP
      const int64_t base = STRIDED ? t0 + gl : t0 + gl * NB;
      float v[NB];
      int c[NB];
      if constexpr (STRIDED) load_entries_strided<NB, G>(a.col, a.val, base, end, c, v);
      else load_entries<NB>(a.col, a.val, base, nnz, c, v);
      if constexpr (STAGED) {
        float y[NB][F];
#pragma unroll
        for (int u = 0; u < NB; ++u)
#pragma unroll
          for (int f = 0; f < F; f += 2) {
            const float2 q = *reinterpret_cast<const float2*>(l12_stage + c[u] * F + f);
            y[u][f] = q.x;
            y[u][f + 1] = q.y;
          }
#pragma unroll
        for (int u = 0; u < NB; ++u)
          if (base + u * STEP < end) {
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = fmaf(v[u], y[u][f], acc[f]);
          }
      } else {
        float2 hv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) hv[u] = *reinterpret_cast<const float2*>(a.H + (xoff + c[u]) * KI);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          if (base + u * STEP < end) {                                    // positions past the row's end: no layer-1 work for them
            const float h[KI] = {hv[u].x, hv[u].y};
            float y[F];
            layer1_row<KI, F>(h, W1, act1, y);
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = fmaf(v[u], y[f], acc[f]);
          }
        }
      }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] += __shfl_xor(acc[f], o);
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
}

// ---- entry-major forward (sparse, possibly skewed rows: the unstaged one-lane-per-row regime) --------------------------
// A block owns 256 consecutive rows; their non-zeros are ONE contiguous range of col / val.  The block walks that range
// entry by entry — thread t takes entries t, t + 256, … of a tile of kEmTile — gathers each entry's H row, applies layer 1
// and parks (val, y[F]) in LDS; then thread t adds up the entries of ITS row from LDS, in entry order (the fmaf chain of
// the one-lane-per-row kernel: the same bits).  Against one lane per row:
//   * three dependent round trips per block whatever the rows look like (row pointers; col / val, coalesced; H rows),
//     instead of two per trip of four non-zeros of the longest row of a wavefront — the reference's chess data has
//     3.97 non-zeros per row and 14.9 in the longest of 64 neighbours: 57 us for a forward that took 22 on the synthetic
//     shape of the same size;
//   * the layer-1 arithmetic is spread evenly over the lanes (per entry, not per row slot).
// Tiles of a block follow each other with the next tile's (col, val) already in flight.  Captured steps, kernel durations
// under rocprofv3: chess 57 -> 37.6 us, the synthetic Bitcoin-OTC shape 21.7 -> 18.2.
// Rows of two slices may share a block (N >= 256: at most one boundary): an entry's H row is found by comparing its
// position with the first entry of the second slice.
#ifndef TMGCN_L12_LONG_TRIPS
#define TMGCN_L12_LONG_TRIPS 8     // trips of a row's G lanes beyond which the whole wave gathers it (l12_bwd_kernel; 16: S2z2 backward 90 us)
#endif
constexpr int kEmTile = 1024;      // 512: chess 56.6 us, 2 048: the synthetic shape 26.7 (two blocks per CU)
#ifndef TMGCN_L12_EM_LONG
#define TMGCN_L12_EM_LONG 64
#endif
#ifndef TMGCN_L12_EM_UNROLL
#define TMGCN_L12_EM_UNROLL 4
#endif
constexpr int kEmLong = TMGCN_L12_EM_LONG;      // entries of one row inside a tile beyond which the row's WAVE sums them (below)
constexpr int kEmUnroll = TMGCN_L12_EM_UNROLL;

// Which row of its block thread t owns (sums from LDS, finishes).  A full block: row t.  A block the caller's partition cut
// short — few rows holding a tile of entries, so long ones — deals its rows over the four waves (row 4·lane + wave), so that
// several hub rows of one block are summed by different waves instead of one after the other by wave 0 (the Reddit-LP shape
// with Zipf sources, traced backward: 45.1 -> 43.4 us; the reference's chess data, rows of at most 83 entries: unchanged).
__device__ __forceinline__ int em_row_of_thread(int t, int rows) { return rows > 192 ? t : ((t & 63) << 2) + (t >> 6); }

// Thread t adds the entries [lo, hi) of ITS row that lie in the tile parked in LDS (value in plane 0, the W planes behind
// it), in entry order.  A segment of more than kEmLong entries — a hub row: one lane walking 1 024 entries is 50 us per tile
// while 255 lanes wait (the Reddit-LP shape with Zipf sources: forward 329 us) — is summed by all 64 lanes of the wave that
// owns the row instead (lanes stride over the entries, butterfly, the owner adds the sum): fixed order, reproducible.
template <int W>
__device__ __forceinline__ void em_row_sum(const float (&park)[1 + W][kEmTile], int tile, int lo, int hi, float (&acc)[W]) {
  const bool seg_long = hi - lo > kEmLong;
  if (!seg_long) {
    // kEmUnroll entries' LDS reads in flight at a time (a lane walking its row one entry per LDS round trip is what the
    // densest row of a wave costs: tools/l12_trace.py, chess — 10 us of a heavy row block's 20); the fmaf chain keeps its order
    int e = lo;
    for (; e + kEmUnroll <= hi; e += kEmUnroll) {
      float w[kEmUnroll], x[kEmUnroll][W];
#pragma unroll
      for (int u = 0; u < kEmUnroll; ++u) {
        w[u] = park[0][e + u - tile];
#pragma unroll
        for (int f = 0; f < W; ++f) x[u][f] = park[1 + f][e + u - tile];
      }
#pragma unroll
      for (int u = 0; u < kEmUnroll; ++u)
#pragma unroll
        for (int f = 0; f < W; ++f) acc[f] = fmaf(w[u], x[u][f], acc[f]);
    }
    for (; e < hi; ++e) {
      const float w = park[0][e - tile];
#pragma unroll
      for (int f = 0; f < W; ++f) acc[f] = fmaf(w, park[1 + f][e - tile], acc[f]);
    }
  }
  const int lane = threadIdx.x & 63;
  for (uint64_t m = __ballot(seg_long); m; m &= m - 1) {
    const int src = __builtin_ctzll(m);
    const int slo = __builtin_amdgcn_readlane(lo, src), shi = __builtin_amdgcn_readlane(hi, src);
    float part[W];
#pragma unroll
    for (int f = 0; f < W; ++f) part[f] = 0.f;
    for (int e = slo + lane; e < shi; e += 64) {
      const float w = park[0][e - tile];
#pragma unroll
      for (int f = 0; f < W; ++f) part[f] = fmaf(w, park[1 + f][e - tile], part[f]);
    }
#pragma unroll
    for (int f = 0; f < W; ++f) part[f] = wave_sum_f32(part[f]);
    if (lane == src) {
#pragma unroll
      for (int f = 0; f < W; ++f) acc[f] += part[f];
    }
  }
}

template <int KI, int F, int NT>
__global__ __launch_bounds__(256) void l12_fwd_em_kernel(L12Args a) {
  static_assert(KI == 2, "H rows are float2");
  __shared__ int64_t rp[257];
  __shared__ float park[1 + F][kEmTile];             // val, y[0..F): one plane each (neighbouring lanes, neighbouring banks)
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
  const int t = threadIdx.x;
  // (one block per row block, the hardware as the scheduler: persistent blocks walking the list with a stride and the next
  // row block's pointers prefetched were measured — chess 27.8 -> 32.9 us, the Zipf Reddit-LP shape 22.2 -> 29.2, S1 18.2 ->
  // 19.4: a block's three or four row blocks in a row balance worse than 3 700 blocks on 1 280 slots; tools/l12_trace.py --fwd)
  // the block's rows: 256 consecutive ones, or — with a partition (tmgcn_layer12_fwd_f32's row_blocks: row blocks cut so
  // that none holds more than about a tile of entries, the heaviest first) — blk[2b + 1] rows from row blk[2b].  (Handing
  // each XCD a run of neighbouring row blocks, as the backward does, brought the forward's fabric fetch from 2.3x the
  // algorithmic bytes to 1.03x — 53.9 -> 23.9 MB on the Bitcoin-OTC shape — and its time nowhere: 18.1 us both ways.)
  L12_STAMP_FWD(0);
  const int64_t first = a.blk ? a.blk[2 * (int64_t)blockIdx.x] : (int64_t)blockIdx.x * 256;
  const int rows = a.blk ? (int)a.blk[2 * (int64_t)blockIdx.x + 1] : (a.n_rows - first < 256 ? (int)(a.n_rows - first) : 256);
  const int rt = em_row_of_thread(t, rows);         // the row of the block this thread sums and finishes
  const int64_t r = first + rt;
  {
    const int64_t q = first + (t < rows ? t : rows);
    rp[t] = a.rowptr[q];
    if (t == 0) rp[256] = a.rowptr[first + rows];
  }
  __syncthreads();
  L12_STAMP_FWD(1);
  const int64_t base = rp[0];
  const int n_ent = (int)(rp[rows < 256 ? rows : 256] - base);
  // the slice boundary inside the block, as an entry position
  const int64_t slice0 = first / a.N;
  const int64_t next_first = (slice0 + 1) * a.N;                                  // first row of the next slice
  const int split = next_first - first < rows ? (int)(rp[next_first - first] - base) : n_ent;
  const int64_t xoff0 = slice0 * a.N;
  const int my_lo = (int)(rp[rt < rows ? rt : rows] - base), my_hi = (int)(rp[rt < rows ? rt + 1 : rows] - base);
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  constexpr int PER = kEmTile / 256;
  int c[PER], c_next[PER];
  float v[PER], v_next[PER];
  auto load_cv = [&](int tile, int (&cc)[PER], float (&vv)[PER]) {   // clamped, unconditional loads: coalesced along the entry range
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tile + u * 256 + t;
      const int64_t q = base + (e < n_ent ? e : (n_ent > 0 ? n_ent - 1 : 0));
      cc[u] = a.col[q];
      vv[u] = a.val[q];
    }
  };
  if (n_ent > 0) load_cv(0, c, v);
#ifdef TMGCN_L12_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  L12_STAMP_FWD(2);
#endif
  for (int tile = 0; tile < n_ent; tile += kEmTile) {
    float2 hv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tile + u * 256 + t;
      // (positions past the block's last entry were clamped to it: they must take ITS slice — the next one may not exist)
      hv[u] = *reinterpret_cast<const float2*>(a.H + (((e < n_ent ? e : n_ent - 1) < split ? xoff0 : xoff0 + a.N) + c[u]) * KI);
    }
    if (tile + kEmTile < n_ent) load_cv(tile + kEmTile, c_next, v_next);      // the next tile's stream, behind this tile's gathers
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tile + u * 256 + t;
      if (e < n_ent) {
        const float h[KI] = {hv[u].x, hv[u].y};
        float y[F];
        layer1_row<KI, F>(h, W1, act1, y);
        park[0][u * 256 + t] = v[u];
#pragma unroll
        for (int f = 0; f < F; ++f) park[1 + f][u * 256 + t] = y[f];
      }
    }
    __syncthreads();
    if (tile == 0) L12_STAMP_FWD(3);
    const int lo = my_lo > tile ? my_lo : tile, hi = my_hi < tile + kEmTile ? my_hi : tile + kEmTile;
    em_row_sum<F>(park, tile, lo, hi, acc);
    __syncthreads();
    if (tile == 0) L12_STAMP_FWD(4);
#pragma unroll
    for (int u = 0; u < PER; ++u) c[u] = c_next[u], v[u] = v_next[u];
  }
  L12_STAMP_FWD(5);
  if (rt < rows) {
    if (a.AX) {
#pragma unroll
      for (int f = 0; f < F; ++f) a.AX[r * F + f] = acc[f];
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float s2 = 0.f;
#pragma unroll
      for (int f = 0; f < F; ++f) s2 = fmaf(acc[f], W2[f][n], s2);
      if (a.pre2_out) a.pre2_out[r * NT + n] = s2;
      a.Z[r * NT + n] = act2(s2);
    }
  }
  L12_STAMP_FWD(14);
}

// The end of a backward block: its dW1 partial sums (NO values, dealt over the G lanes of a row group: lane gl owns
// q = gl + j·G) folded over the block in a fixed order, stored write-through as the block's slab; the block that draws the
// last ticket adds all slabs in order and writes dW1.
template <int NO, int G>
__device__ __forceinline__ void l12_bwd_finish(const double (&acc)[(NO + G - 1) / G], const L12Args& a) {
  constexpr int NPL = (NO + G - 1) / G;
  __shared__ double red[4][NO];
  const int gl = threadIdx.x & (G - 1);
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
  __shared__ double total[NO];
  if (!slab_tree_finish<NO>(reinterpret_cast<unsigned*>(a.part), (int)gridDim.x, a.sync, total)) return;
  if (threadIdx.x < NO) a.dW1[threadIdx.x] = (float)total[threadIdx.x];
}

// Backward.  Groups of G lanes walk rows r = group, group + n_groups, …; the KI·F fp64 accumulators of dW1 are dealt
// over the lanes of a group (lane gl owns q = gl + j·G).
static_assert(kSyncGroups == TMGCN_L12_RUNS, "include/tmgcn.h states the number of runs a row-block list is read as");
constexpr int kL12MaxBlocks = 4096;          // slabs the workspace holds
constexpr int kL12ResidentBlocks = 1024;     // a persistent grid: four blocks per CU

template <int KI, int F, int NT, int G, bool STAGED, bool ACT2>
__global__ __launch_bounds__(256) void l12_bwd_kernel(L12Args a) {
  constexpr int NO = KI * F;
  constexpr int NPL = (NO + G - 1) / G;
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
  const int64_t nnz = a.rowptr[a.n_rows];
  const int gl = threadIdx.x & (G - 1);
  double acc[NPL];
#pragma unroll
  for (int j = 0; j < NPL; ++j) acc[j] = 0.0;
  int64_t r, r_end, r_step;
  int64_t r_base = 0;
  if constexpr (STAGED) {
    const int slice = blockIdx.x / a.chunks, chunk = blockIdx.x - slice * a.chunks;
    const int64_t xoff = (int64_t)slice * a.N;
    for (int c = threadIdx.x; c < a.N; c += 256) {               // dZ ⊙ act2'(pre2) of every node of the slice, once
      const float2* gz = reinterpret_cast<const float2*>(a.dZ + (xoff + c) * NT);
      const float2* pz = reinterpret_cast<const float2*>((ACT2 ? a.pre2 : a.dZ) + (xoff + c) * NT);
#pragma unroll
      for (int i = 0; i < NT / 2; ++i) {
        float2 q = gz[i];
        if constexpr (ACT2) {
          const float2 w = pz[i];
          q.x *= dact2(w.x);
          q.y *= dact2(w.y);
        }
        *reinterpret_cast<float2*>(l12_stage + c * NT + 2 * i) = q;
      }
    }
    const int first = chunk * a.chunk_rows, last = first + a.chunk_rows < a.N ? first + a.chunk_rows : a.N;
    int64_t* rps = reinterpret_cast<int64_t*>(l12_stage + a.N * NT);      // the block's row pointers
    for (int i = threadIdx.x; i <= last - first; i += 256) rps[i] = a.rowptr[xoff + first + i];
    r_base = xoff + first;
    __syncthreads();
    r = xoff + first + threadIdx.x / G;
    r_end = xoff + last;
    r_step = 256 / G;
  } else {
    r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    r_end = a.n_rows;
    r_step = (int64_t)gridDim.x * 256 / G;
  }
  constexpr int NB = STAGED ? 8 : 4;                         // NB consecutive non-zeros per lane and trip (see the forward kernel)
  // unstaged: a row that would take its G lanes more than kL12LongTrips trips (a hub: 3 800 entries on two lanes are 470
  // trips, 200 us of a launch that takes 30 without it) is gathered by all 64 lanes of the wave after the groups' own rows
  constexpr bool WAVE_ROWS = !STAGED && G < 64;
  constexpr int kL12LongTrips = TMGCN_L12_LONG_TRIPS;
  for (;; r += r_step) {
    const bool live = r < r_end;
    if constexpr (WAVE_ROWS) {
      if (__ballot(live) == 0) break;                      // (the wave leaves together: its long rows need all lanes)
    } else {
      if (!live) break;
    }
    int64_t beg = 0, end = 0;
    if constexpr (STAGED) {
      const int64_t* rps = reinterpret_cast<const int64_t*>(l12_stage + a.N * NT);
      beg = rps[r - r_base], end = rps[r - r_base + 1];
    } else if (live) {
      beg = a.rowptr[r], end = a.rowptr[r + 1];
    }
    const int64_t xoff = STAGED ? 0 : (r / a.N) * (int64_t)a.N;   // staged: columns index the slice's LDS copy
    float t[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) t[n] = 0.f;
    const bool is_long = WAVE_ROWS && end - beg > (int64_t)kL12LongTrips * NB * G;
    for (int64_t t0 = beg; t0 < (is_long ? beg : end); t0 += NB * G) {
      constexpr bool STRIDED = !STAGED && G > 1;
      constexpr int STEP = STRIDED ? G : 1;                   // entry of slot u: base + u·STEP
      const int64_t base = STRIDED ? t0 + gl : t0 + gl * NB;
      float v[NB];
      int c[NB];
      if constexpr (STRIDED) load_entries_strided<NB, G>(a.col, a.val, base, end, c, v);
      else load_entries<NB>(a.col, a.val, base, nnz, c, v);
      float g[NB][NT];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const float2* gz = STAGED ? reinterpret_cast<const float2*>(l12_stage + c[u] * NT)
                                  : reinterpret_cast<const float2*>(a.dZ + (xoff + c[u]) * NT);
#pragma unroll
        for (int i = 0; i < NT / 2; ++i) {
          const float2 q = gz[i];
          g[u][2 * i] = q.x;
          g[u][2 * i + 1] = q.y;
        }
      }
      if constexpr (!STAGED && ACT2) {                                // act2 != none: dZ ⊙ act2'(pre2) of the gathered rows
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const float2* pz = reinterpret_cast<const float2*>(a.pre2 + (xoff + c[u]) * NT);
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
        if (base + u * STEP < end) {
#pragma unroll
          for (int n = 0; n < NT; ++n) t[n] = fmaf(v[u], g[u][n], t[n]);
        }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int n = 0; n < NT; ++n) t[n] += __shfl_xor(t[n], o);
    if constexpr (WAVE_ROWS) {
      const int lane = threadIdx.x & 63;
      for (uint64_t m = __ballot(is_long && gl == 0); m; m &= m - 1) {
        const int src = __builtin_ctzll(m);
        const int64_t r2 = l12_readlane64(r, src), b2 = l12_readlane64(beg, src), e2 = l12_readlane64(end, src);
        const int64_t xoff2 = (r2 / a.N) * (int64_t)a.N;
        float tw[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) tw[n] = 0.f;
        for (int64_t p = b2 + lane; p < e2; p += NB * 64) {
          int c[NB];
          float v[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) {                    // clamped, unconditional loads; masked by a zero value
            const int64_t q = p + u * 64;
            const int64_t qc = q < e2 ? q : e2 - 1;
            c[u] = a.col[qc];
            v[u] = q < e2 ? a.val[qc] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < NB; ++u) {
            const float2* gz = reinterpret_cast<const float2*>(a.dZ + (xoff2 + c[u]) * NT);
            const float2* pz = reinterpret_cast<const float2*>((ACT2 ? a.pre2 : a.dZ) + (xoff2 + c[u]) * NT);
#pragma unroll
            for (int i = 0; i < NT / 2; ++i) {
              float2 q = gz[i];
              if constexpr (ACT2) {
                const float2 w = pz[i];
                q.x *= dact2(w.x);
                q.y *= dact2(w.y);
              }
              tw[2 * i] = fmaf(v[u], q.x, tw[2 * i]);
              tw[2 * i + 1] = fmaf(v[u], q.y, tw[2 * i + 1]);
            }
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
          for (int n = 0; n < NT; ++n) tw[n] += __shfl_xor(tw[n], o);
        if (lane / G == src / G) {                          // the row's own group goes on with the sum
#pragma unroll
          for (int n = 0; n < NT; ++n) t[n] = tw[n];
        }
      }
      if (!live) continue;
    }
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
  l12_bwd_finish<NO, G>(acc, a);
}

// ---- entry-major backward (the counterpart of l12_fwd_em_kernel, on the transposed CSR) ---------------------------------
// Persistent blocks take row blocks rb = blockIdx, blockIdx + gridDim, …; a row block's entries are walked tile by tile —
// each entry's dZ row (⊙ act2'(pre2) when layer 2 has an activation) parked in LDS with its value —, thread t sums ITS
// row in entry order (the one-lane-per-row chain: the same bits), finishes the row (·W2ᵀ, act1', H[r]ᵀ·dP into its fp64
// dW1 partial sums) and the block ends like every backward block (l12_bwd_finish).
// DW2: the kernel also forms dW2 = AXᵀ·(dZ ⊙ act2'(pre2)) — layer 2's weight gradient, which needs nothing but the rows of
// two [R][6] tensors: at the start of a row block, while the row pointers are in flight, four lanes per row (the narrow dW
// kernel's 3 x 3 quadrants, gemm.hip) load the row block's own rows of AX and dZ, multiply and fold over the wave (two DPP
// row rotations, two shuffles); the sums join the row block's slab behind the dW1 ones.  A launch of its own for these
// 28 MB cost 10.3 us of every step (5 of them its ticket tail).
template <int KI, int F, int NT, bool ACT2, bool DW2>
__global__ __launch_bounds__(256) void l12_bwd_em_kernel(L12Args a) {
  constexpr int NO1 = KI * F, NO = NO1 + (DW2 ? F * NT : 0);
  __shared__ int64_t rp[257];
  __shared__ float park[1 + NT][kEmTile];            // val, g[0..NT)
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
  const int t = threadIdx.x;
  constexpr int PER = kEmTile / 256;
  const int64_t n_row_blocks = a.blk ? a.n_blk : (a.n_rows + 255) / 256;
  __shared__ double red[4][NO];
  L12_STAMP(0);
  [[maybe_unused]] int trace_it = 0;
  // Row blocks are DRAWN, not dealt: the blocks of hand-off group g (blockIdx ≡ g mod 16, common.h) share a run of the list
  // (below) — a block's first one by its place in the group, every further one from the group's counter (an int of the
  // launch's hand-off block: sixteen addresses, agent-scope atomics on one are served one at a time), drawn while the
  // current row block is being worked on.  A block that waits on a heavy row block no longer holds others back, and the
  // ticket tail (4.6 us of latency, tools/l12_trace.py) is paid once per resident block instead of once per row block.
  // The dW1 partial sums are kept per ROW BLOCK (slab rb: the rows' H[r]ᵀ·dP[r] folded over the block in a fixed order), not
  // per executing block: which block draws which row block varies from run to run, the slabs and the order they are added
  // in (slab_tree_finish_in: group g adds the slabs of its run, then the sixteen group sums) do not — bit-reproducible.
  __shared__ int64_t s_next;
  const int grp = blockIdx.x % kSyncGroups;
  const int64_t grp_members = ((int64_t)gridDim.x - grp + kSyncGroups - 1) / kSyncGroups;
  int* draw = a.sync + (1 + grp) * kSyncStride + 8;
  // (group g — the thread blocks of one XCD, as long as blocks are dealt to the XCDs in turn — works on the run of row blocks
  // [n·g / groups, n·(g + 1) / groups) of the list: neighbouring row blocks gather the same rows of dZ, and an XCD's L2 that
  // sees a slice alone fetches its lines once instead of every XCD fetching them — fabric fetch of the Bitcoin-OTC-shaped
  // backward 126 -> 53 MB, 31.2 -> 27.8 us; with row blocks ≡ g mod 16 every XCD touched every slice.  Runs of equal LENGTH:
  // runs of equal estimated work, given by the caller, were measured too — the Zipf Reddit-LP shape 40.7 us against 38.4,
  // the reference's chess data 48.1 against 48.9, where row blocks ≡ g mod 16 of one heaviest-first list ran 42.5 / 45.9 — all at
  // five blocks per CU; at four: runs 38.2-39.1 / 45.9, ≡ g mod 16 38.7-39.3 / 45.0)
  const int n_grp = gridDim.x < kSyncGroups ? (int)gridDim.x : kSyncGroups;
  const int64_t run_lo = n_row_blocks * grp / n_grp, run_hi = n_row_blocks * (grp + 1) / n_grp;
  const int64_t run_first = run_lo + blockIdx.x / kSyncGroups;
  for (int64_t rb = run_first < run_hi ? run_first : n_row_blocks; rb < n_row_blocks;) {
    const int64_t first = a.blk ? a.blk[2 * rb] : rb * 256;
    const int rows = a.blk ? (int)a.blk[2 * rb + 1] : (a.n_rows - first < 256 ? (int)(a.n_rows - first) : 256);
    const int rt = em_row_of_thread(t, rows);
    const int64_t r = first + rt;
    rp[t] = a.rowptr[first + (t < rows ? t : rows)];   // (the previous row block's readers of rp are done: the barrier at the loop's end)
    if (t == 0) rp[256] = a.rowptr[first + rows];
    const float2 hv = *reinterpret_cast<const float2*>(a.H + (rt < rows ? r : first) * KI);   // this row's H, early
    constexpr int KH = F / 2, NH = NT / 2;
    [[maybe_unused]] float qx[4][KH], qg[4][NH];
    if constexpr (DW2) {                              // rows q, q + 64, … of the row block: lane j = t & 3 holds quadrant (j >> 1, j & 1)
      const int q = t >> 2, kh = (t >> 1) & 1, nh = t & 1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool in = q + 64 * u < rows;
        const int64_t row = first + (in ? q + 64 * u : 0);
#pragma unroll
        for (int i = 0; i < KH; ++i) qx[u][i] = a.AX[row * F + kh * KH + i];
#pragma unroll
        for (int i = 0; i < NH; ++i) {
          float gq = a.dZ[row * NT + nh * NH + i];
          if constexpr (ACT2) gq *= dact2(a.pre2[row * NT + nh * NH + i]);
          qg[u][i] = in ? gq : 0.f;
        }
      }
    }
    __syncthreads();
    if (trace_it < 3) L12_STAMP(1 + 4 * trace_it);
    const int64_t base = rp[0];
    const int n_ent = (int)(rp[rows] - base);
    const int64_t slice0 = first / a.N;
    const int64_t next_first = (slice0 + 1) * a.N;
    const int split = next_first - first < rows ? (int)(rp[next_first - first] - base) : n_ent;
    const int64_t xoff0 = slice0 * a.N;
    const int my_lo = (int)(rp[rt < rows ? rt : rows] - base), my_hi = (int)(rp[rt < rows ? rt + 1 : rows] - base);
    float ts[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) ts[n] = 0.f;
    int c[PER], c_next[PER];
    float v[PER], v_next[PER];
    auto load_cv = [&](int tile, int (&cc)[PER], float (&vv)[PER]) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int e = tile + u * 256 + t;
        const int64_t q = base + (e < n_ent ? e : (n_ent > 0 ? n_ent - 1 : 0));
        cc[u] = a.col[q];
        vv[u] = a.val[q];
      }
    };
    if (n_ent > 0) load_cv(0, c, v);
    int drawn = 0;
    if (t == 0) drawn = __hip_atomic_fetch_add(draw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // behind the (col, val) loads: back with the gathers
    // (dW2's sums while the first (col, val) tile is in flight: its rows were asked for ahead of the row pointers' barrier)
    if constexpr (DW2) {
      const int lane = t & 63, wave = t >> 6, kh = (t >> 1) & 1, nh = t & 1;
#pragma unroll
      for (int k = 0; k < KH; ++k)
#pragma unroll
        for (int n = 0; n < NH; ++n) {
          // the 64 rows of a wave in fp32 (a dot product of 64 terms: 4 rows per lane, 16 lanes per quadrant), fp64 from there
          // on — across the waves, the row blocks and the groups; in fp64 throughout (conversions, half-rate FMAs, three
          // instructions per DPP step) this was 5 us of vector-ALU time per launch at 4-5 waves per SIMD, not latency
          float w = 0.f;
#pragma unroll
          for (int u = 0; u < 4; ++u) w = fmaf(qx[u][k], qg[u][n], w);
          w += dpp_lane_f32<0x124>(w);                // row_ror:4, row_ror:8: the four lanes of a 16-lane row with this quadrant
          w += dpp_lane_f32<0x128>(w);
          w += __shfl_xor(w, 16);
          w += __shfl_xor(w, 32);
          if (lane < 4) red[wave][NO1 + (kh * KH + k) * NT + nh * NH + n] = (double)w;
        }
    }
#ifdef TMGCN_L12_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (trace_it < 3) L12_STAMP(2 + 4 * trace_it);
#endif
    for (int tile = 0; tile < n_ent; tile += kEmTile) {
      float g[PER][NT];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int e = tile + u * 256 + t;
        const int64_t row = ((e < n_ent ? e : n_ent - 1) < split ? xoff0 : xoff0 + a.N) + c[u];   // clamped positions: the last entry's slice
        const float2* gz = reinterpret_cast<const float2*>(a.dZ + row * NT);
#pragma unroll
        for (int i = 0; i < NT / 2; ++i) {
          const float2 q = gz[i];
          g[u][2 * i] = q.x;
          g[u][2 * i + 1] = q.y;
        }
        if constexpr (ACT2) {
          const float2* pz = reinterpret_cast<const float2*>(a.pre2 + row * NT);
#pragma unroll
          for (int i = 0; i < NT / 2; ++i) {
            const float2 q = pz[i];
            g[u][2 * i] *= dact2(q.x);
            g[u][2 * i + 1] *= dact2(q.y);
          }
        }
      }
      if (tile + kEmTile < n_ent) load_cv(tile + kEmTile, c_next, v_next);
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int e = tile + u * 256 + t;
        if (e < n_ent) {
          park[0][u * 256 + t] = v[u];
#pragma unroll
          for (int n = 0; n < NT; ++n) park[1 + n][u * 256 + t] = g[u][n];
        }
      }
      __syncthreads();
      if (tile == 0 && trace_it < 3) L12_STAMP(3 + 4 * trace_it);
      const int lo = my_lo > tile ? my_lo : tile, hi = my_hi < tile + kEmTile ? my_hi : tile + kEmTile;
      em_row_sum<NT>(park, tile, lo, hi, ts);
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PER; ++u) c[u] = c_next[u], v[u] = v_next[u];
    }
    if (trace_it < 3) L12_STAMP(4 + 4 * trace_it);
    ++trace_it;
    if (t == 0) s_next = run_lo + grp_members + drawn < run_hi ? run_lo + grp_members + drawn : n_row_blocks;
    {
      // dY = t·W2ᵀ, P = H·W1, dP = dY ⊙ act1'(P): as l12_bwd_kernel; the row's share of dW1 (zero for a thread without a row)
      // summed over the wave, the four waves' sums over the block
      const float h[KI] = {hv.x, hv.y};
      const int lane = t & 63, wave = t >> 6;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        float sy = 0.f;
#pragma unroll
        for (int n = 0; n < NT; ++n) sy = fmaf(ts[n], W2[f][n], sy);
        float pf = 0.f;
#pragma unroll
        for (int k = 0; k < KI; ++k) pf = fmaf(h[k], W1[k][f], pf);
        const float dP = rt < rows ? sy * dact1(pf) : 0.f;
#pragma unroll
        for (int k = 0; k < KI; ++k) {
          const double w = wave_sum_f64((double)h[k] * (double)dP);
          if (lane == 0) red[wave][k * F + f] = w;
        }
      }
    }
    __syncthreads();
    if (t < NO)
      __hip_atomic_store(reinterpret_cast<unsigned*>(a.part) + rb * NO + t,
                         __float_as_uint((float)(((red[0][t] + red[1][t]) + red[2][t]) + red[3][t])), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    rb = s_next;
  }
  L12_STAMP(13);
  __shared__ double total[NO];
  static_assert(sizeof(park) >= sizeof(double) * (256 / NO) * NO, "the tile planes double as the finisher's partial sums");
  if (slab_tree_finish_in<NO>(reinterpret_cast<unsigned*>(a.part), (int)gridDim.x, a.sync, total, (int)n_row_blocks,
                              reinterpret_cast<double*>(&park[0][0]), true)) {      // (park: every reader passed the loop's last barrier)
    // (the last block of all: every other block drew its last row block before it took its ticket — the draw counters go
    // back to zero with the tickets)
    if (t < kSyncGroups) __hip_atomic_store(a.sync + (1 + t) * kSyncStride + 8, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < NO1) a.dW1[t] = (float)total[t];
    else if (t < NO) a.dW2[t - NO1] = (float)total[t];
  }
  L12_STAMP(14);
}

// lanes per row: chains of about four non-zeros per lane, walked NB at a time (kernel durations under rocprofv3, captured
// S1 / S3 steps: 3 nnz/row: G = 1 beats 2 by 18 %; 27 nnz/row: G = 4 — 41 us backward against 48 with one lane per row)
static int l12_lanes(float avg_nnz_per_row, bool staged) {
  int G = 8;
  if (avg_nnz_per_row >= 0.f) {
    G = 1;
    if (staged)       // 8 per lane and trip, two trips: 27 nnz/row: G = 2 21.8 / 23.2 us, 4: 25.9 / 24.0 (and 32 B of scratch), 8: 31.2 / 30.8
      while (G < 16 && 16.f * G < avg_nnz_per_row) G <<= 1;
    else
      while (G < 16 && 4.f * G <= avg_nnz_per_row) G <<= 1;
  }
  return G;
}

// blocks per slice: about 640 blocks in all (2.5 per CU; every block forms the whole slice in LDS first, so more blocks
// means more of that), at least 64 rows each.  Captured AMLSim-shaped step (150 slices of 1 000 nodes, 27 per row), kernel
// durations under rocprofv3: 4 blocks per slice 21.8 us forward / 23.2 backward, 6: 21.8 / 27.8, 2: 33.8 / 33.4.
static int l12_chunks(int64_t slices, int32_t N) {
  int64_t c = 640 / slices, most = (N + 63) / 64;
  if (c > most) c = most;
  return (int)(c < 1 ? 1 : c);
}

// The staged variants (see the head of the file): LDS for one slice, at least 8 non-zeros per row to pay for forming it.
// Everything the launch asks for has to fit the 64 KB a block gets without hipFuncSetAttribute: the slice ([N][width]
// floats), the block's row-pointer slab ((chunk_rows + 1) int64) and the kernels' static arrays (the backward's fp64
// reduction scratch; 4 KB covers every instantiation).  A slice that does not fit takes the unstaged / entry-major kernels.
constexpr int64_t kL12LdsLimit = 64 << 10, kL12StaticLds = 4 << 10;
static bool l12_staged(int64_t n_rows, int32_t N, int width, float avg_nnz_per_row) {
  if (!(avg_nnz_per_row >= 8.f) || n_rows / N > kL12MaxBlocks) return false;
  const int chunks = l12_chunks(n_rows / N, N);
  const int64_t chunk_rows = (N + chunks - 1) / chunks;
  return (int64_t)N * width * 4 + (chunk_rows + 1) * 8 + kL12StaticLds <= kL12LdsLimit;
}

template <int F, int NT, bool BWD, bool STAGED>
static void l12_launch_g(const L12Args& a, int G, unsigned blocks, size_t lds, hipStream_t st) {
#define TMGCN_L12(G_)                                                                                              \
  if (BWD) {                                                                                                       \
    if (a.pre2) hipLaunchKernelGGL((l12_bwd_kernel<2, F, NT, G_, STAGED, true>), dim3(blocks), dim3(256), lds, st, a);  \
    else hipLaunchKernelGGL((l12_bwd_kernel<2, F, NT, G_, STAGED, false>), dim3(blocks), dim3(256), lds, st, a);        \
  }                                                                                                                \
  else hipLaunchKernelGGL((l12_fwd_kernel<2, F, NT, G_, STAGED>), dim3(blocks), dim3(256), lds, st, a);
  switch (G) {
    case 1: TMGCN_L12(1) break;
    case 2: TMGCN_L12(2) break;
    case 4: TMGCN_L12(4) break;
    case 8: TMGCN_L12(8) break;
    default: TMGCN_L12(16)
  }
#undef TMGCN_L12
}

template <int F>
static void l12_em_launch_n(const L12Args& a, int NT, unsigned blocks, hipStream_t st) {
  switch (NT) {
    case 2: hipLaunchKernelGGL((l12_fwd_em_kernel<2, F, 2>), dim3(blocks), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((l12_fwd_em_kernel<2, F, 4>), dim3(blocks), dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL((l12_fwd_em_kernel<2, F, 6>), dim3(blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((l12_fwd_em_kernel<2, F, 8>), dim3(blocks), dim3(256), 0, st, a);
  }
}
static void l12_em_launch(const L12Args& a, int F, int NT, unsigned blocks, hipStream_t st) {
  switch (F) {
    case 2: l12_em_launch_n<2>(a, NT, blocks, st); break;
    case 4: l12_em_launch_n<4>(a, NT, blocks, st); break;
    case 6: l12_em_launch_n<6>(a, NT, blocks, st); break;
    default: l12_em_launch_n<8>(a, NT, blocks, st);
  }
}

template <int F, int NT, bool ACT2, bool DW2>
static void l12_bwd_em_launch_d(const L12Args& a, int64_t row_blocks, hipStream_t st) {
  // (with dW2 inside, four blocks per CU beat the five that fit: chess 46.0 against 48.6 us, S1 26.5 / 28.3, the Zipf Reddit-LP
  // shape 38.0 / 40.4; three: 52.3 / 30.8 / 42.3 — without dW2 five and four were 37.1 / 37.8 on chess)
  const int64_t resident = persistent_grid(l12_bwd_em_kernel<2, F, NT, ACT2, DW2>, 256, 0, DW2 ? 4 : 5);
  hipLaunchKernelGGL((l12_bwd_em_kernel<2, F, NT, ACT2, DW2>), dim3((unsigned)(row_blocks < resident ? row_blocks : resident)), dim3(256), 0, st, a);
}
template <int F, int NT, bool ACT2>
static void l12_bwd_em_launch_g(const L12Args& a, int64_t row_blocks, hipStream_t st) {
  if (a.dW2) l12_bwd_em_launch_d<F, NT, ACT2, true>(a, row_blocks, st);
  else l12_bwd_em_launch_d<F, NT, ACT2, false>(a, row_blocks, st);
}
template <int F, bool ACT2>
static void l12_bwd_em_launch_n(const L12Args& a, int NT, int64_t row_blocks, hipStream_t st) {
  switch (NT) {
    case 2: l12_bwd_em_launch_g<F, 2, ACT2>(a, row_blocks, st); break;
    case 4: l12_bwd_em_launch_g<F, 4, ACT2>(a, row_blocks, st); break;
    default: l12_bwd_em_launch_g<F, 6, ACT2>(a, row_blocks, st);
  }
}
static void l12_bwd_em_launch(const L12Args& a, int F, int NT, int64_t blocks, hipStream_t st) {
#define TMGCN_EM_B(F_) (a.pre2 ? l12_bwd_em_launch_n<F_, true>(a, NT, blocks, st) : l12_bwd_em_launch_n<F_, false>(a, NT, blocks, st))
  switch (F) {
    case 2: TMGCN_EM_B(2); break;
    case 4: TMGCN_EM_B(4); break;
    default: TMGCN_EM_B(6);
  }
#undef TMGCN_EM_B
}

template <bool BWD, bool STAGED>
static void l12_launch(const L12Args& a, int F, int NT, int G, unsigned blocks, size_t lds, hipStream_t st) {
#define TMGCN_L12_N(F_)                                                             \
  switch (NT) {                                                                     \
    case 2: l12_launch_g<F_, 2, BWD, STAGED>(a, G, blocks, lds, st); break;           \
    case 4: l12_launch_g<F_, 4, BWD, STAGED>(a, G, blocks, lds, st); break;           \
    case 6: l12_launch_g<F_, 6, BWD, STAGED>(a, G, blocks, lds, st); break;           \
    default: l12_launch_g<F_, 8, BWD, STAGED>(a, G, blocks, lds, st);                 \
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
                                      float avg_nnz_per_row, const int64_t* row_blocks, int32_t n_row_blocks, void* stream) {
  TMGCN_REQUIRE(tmgcn_layer12_supported(K0, F, Nf), "layer12: unsupported widths %d -> %d -> %d (2 -> even <= 8 -> even <= 8)", K0, F, Nf);
  TMGCN_REQUIRE(n_rows >= 0 && N > 0 && n_rows % N == 0, "layer12: bad shape n_rows=%lld N=%d", (long long)n_rows, N);
  TMGCN_REQUIRE(act1 >= TMGCN_ACT_NONE && act1 <= TMGCN_ACT_SELU && act2 >= TMGCN_ACT_NONE && act2 <= TMGCN_ACT_SELU,
                "layer12: unknown activation");
  if (n_rows == 0) return TMGCN_OK;
  TMGCN_REQUIRE(rowptr && H && W1 && W2 && Z, "layer12: null pointer");
  TMGCN_REQUIRE(reinterpret_cast<uintptr_t>(H) % 8 == 0, "layer12: H must be 8-byte aligned");
  TMGCN_REQUIRE((row_blocks == nullptr) == (n_row_blocks == 0) && n_row_blocks >= 0, "layer12: row_blocks and n_row_blocks go together");
  L12Args a{rowptr, col, val, H, W1, W2, nullptr, nullptr, Z, AX, pre2, nullptr, nullptr, nullptr, nullptr, n_rows, N, act1, act2, 0, 0, nullptr, 0};
  const bool staged = l12_staged(n_rows, N, F, avg_nnz_per_row);
  const int G = l12_lanes(avg_nnz_per_row, staged);
  if (staged) {
    a.chunks = l12_chunks(n_rows / N, N);
    a.chunk_rows = (N + a.chunks - 1) / a.chunks;
    l12_launch<false, true>(a, F, Nf, G, (unsigned)(n_rows / N * a.chunks), (size_t)N * F * 4 + (a.chunk_rows + 1) * 8, (hipStream_t)stream);
  } else if (N >= 256) {
    a.blk = row_blocks;                                                                  // entry-major: see l12_fwd_em_kernel
    a.n_blk = n_row_blocks;
    l12_em_launch(a, F, Nf, row_blocks ? (unsigned)n_row_blocks : (unsigned)((n_rows + 255) / 256), (hipStream_t)stream);
  } else {
    l12_launch<false, false>(a, F, Nf, G, (unsigned)((n_rows * G + 255) / 256), 0, (hipStream_t)stream);
  }
  return check_launch("layer12_fwd");
}

// 1 when the fused forward is the faster route for this shape: slices of at least 256 nodes (the entry-major kernel:
// layer 1 is applied per ENTRY, evenly over the lanes — 5 / 8 / 16 / 40 non-zeros per row of 8 000 - 20 000 nodes: 24.6 / 33.8 /
// 59.2 / 195.7 us against 46.4 fused with two lanes per row and 50.1 / 72.1 / 234 for the GEMM + fused SpMM pair), a shape
// the staged variant takes (layer 1 once per node), or short rows; otherwise the caller forms act1(H·W1) itself and calls
// tmgcn_spmm_gemm_f32 (the same Z up to the fp32 summation order of a row).
extern "C" int tmgcn_layer12_fwd_pays(int64_t n_rows, int32_t N, int32_t F, float avg_nnz_per_row) {
  if (N <= 0 || n_rows <= 0) return 0;
  return (N >= 256 || (avg_nnz_per_row >= 0.f && avg_nnz_per_row <= 6.f) || l12_staged(n_rows, N, F, avg_nnz_per_row)) ? 1 : 0;
}

extern "C" int64_t tmgcn_layer12_bwd_workspace_bytes(int32_t K0, int32_t F, int32_t Nf, int64_t n_rows, int32_t n_row_blocks) {
  // slabs of the blocks (lanes-per-row and staged kernels: at most kL12MaxBlocks) or of the row blocks (entry-major) + group
  // slabs, each wide enough for dW1 and dW2
  int64_t slabs = n_row_blocks > 0 ? n_row_blocks : (n_rows + 255) / 256;
  if (slabs < kL12MaxBlocks) slabs = kL12MaxBlocks;
  return (slabs + kSyncGroups) * ((int64_t)K0 * F + (int64_t)F * Nf) * (int64_t)sizeof(float);
}

// the walk tmgcn_layer12_bwd_f32 takes (see there)
static bool l12_bwd_entry_major(int64_t n_rows, int32_t N, int32_t F, int32_t Nf, float avg_nnz_per_row, bool partition) {
  const bool staged = l12_staged(n_rows, N, Nf, avg_nnz_per_row);
  return !staged && (l12_lanes(avg_nnz_per_row, staged) == 1 || partition) && N >= 256 && F <= 6 && Nf <= 6;
}

extern "C" int tmgcn_layer12_bwd_forms_dw2(int64_t n_rows, int32_t N, int32_t F, int32_t Nf, float avg_nnz_per_row, int32_t n_row_blocks) {
  return (n_rows > 0 && N > 0 && l12_bwd_entry_major(n_rows, N, F, Nf, avg_nnz_per_row, n_row_blocks > 0)) ? 1 : 0;
}

extern "C" int tmgcn_layer12_bwd_f32(const int64_t* t_rowptr, const int32_t* t_col, const float* t_val, const float* dZ,
                                      const float* pre2, const float* H, const float* W1, int32_t act1, const float* W2,
                                      int32_t act2, int64_t n_rows, int32_t N, int32_t K0, int32_t F, int32_t Nf,
                                      float* dW1, const float* AX, float* dW2, float avg_nnz_per_row, const int64_t* row_blocks,
                                      int32_t n_row_blocks, void* workspace, int64_t workspace_bytes, void* stream) {
  TMGCN_REQUIRE(tmgcn_layer12_supported(K0, F, Nf), "layer12_bwd: unsupported widths %d -> %d -> %d", K0, F, Nf);
  TMGCN_REQUIRE(n_rows > 0 && N > 0 && n_rows % N == 0, "layer12_bwd: bad shape n_rows=%lld N=%d", (long long)n_rows, N);
  TMGCN_REQUIRE(act1 >= TMGCN_ACT_NONE && act1 <= TMGCN_ACT_SELU && act2 >= TMGCN_ACT_NONE && act2 <= TMGCN_ACT_SELU,
                "layer12_bwd: unknown activation");
  TMGCN_REQUIRE(t_rowptr && dZ && H && W1 && W2 && dW1 && workspace, "layer12_bwd: null pointer");
  TMGCN_REQUIRE((act2 == TMGCN_ACT_NONE) == (pre2 == nullptr), "layer12_bwd: pre2 must be given exactly when act2 is not none");
  TMGCN_REQUIRE(reinterpret_cast<uintptr_t>(H) % 8 == 0 && reinterpret_cast<uintptr_t>(dZ) % 8 == 0 &&
                    (!pre2 || reinterpret_cast<uintptr_t>(pre2) % 8 == 0),
                "layer12_bwd: H, dZ and pre2 must be 8-byte aligned");
  TMGCN_REQUIRE((AX == nullptr) == (dW2 == nullptr), "layer12_bwd: AX and dW2 go together");
  TMGCN_REQUIRE(!dW2 || l12_bwd_entry_major(n_rows, N, F, Nf, avg_nnz_per_row, n_row_blocks > 0),
                "layer12_bwd: this call does not take the entry-major kernel and cannot form dW2 (ask tmgcn_layer12_bwd_forms_dw2)");
  if (workspace_bytes < tmgcn_layer12_bwd_workspace_bytes(K0, F, Nf, n_rows, n_row_blocks)) {
    set_error("layer12_bwd: workspace too small");
    return TMGCN_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  TMGCN_REQUIRE((row_blocks == nullptr) == (n_row_blocks == 0) && n_row_blocks >= 0, "layer12_bwd: row_blocks and n_row_blocks go together");
  L12Args a{t_rowptr, t_col, t_val, H, W1, W2, dZ, pre2, nullptr, const_cast<float*>(AX), nullptr, dW1, dW2, (float*)workspace,
            acquire_sync_word(st), n_rows, N, act1, act2, 0, 0, nullptr, 0};
  TMGCN_REQUIRE(a.sync, "layer12_bwd: no hand-off block: %s", pool_error());
  const bool staged = l12_staged(n_rows, N, Nf, avg_nnz_per_row);
  const int G = l12_lanes(avg_nnz_per_row, staged);
  if (staged) {
    a.chunks = l12_chunks(n_rows / N, N);
    a.chunk_rows = (N + a.chunks - 1) / a.chunks;
    l12_launch<true, true>(a, F, Nf, G, (unsigned)(n_rows / N * a.chunks), (size_t)N * Nf * 4 + (a.chunk_rows + 1) * 8, st);
  } else if (l12_bwd_entry_major(n_rows, N, F, Nf, avg_nnz_per_row, row_blocks != nullptr)) {
    // entry-major (l12_bwd_em_kernel): sparse rows (one lane per row otherwise), and whenever the caller brings a partition of
    // the rows — it does for SKEWED adjacencies (ops.layer12: hub rows; the Reddit-LP shape with Zipf sources, 11.8 entries per
    // row and hubs of 3 800: 45.8 us against 80.1 for the lanes-per-row kernel; uniform rows of 8 / 16 / 40 entries: 38 / 53 /
    // 132 against 30 / 62 / 106 — no partition is passed there).  Captured steps, kernel durations under rocprofv3: the synthetic Bitcoin-OTC shape
    // 31.1 -> 27.9 us.  Grid: resident blocks that DRAW their row blocks (see the kernel).  One block per row block with the
    // hardware as the scheduler cost every row block the slab / ticket tail and a block launch (chess: 2 661 blocks of mean
    // life 12.9 us, 4.6 of it the tail: 49.0 us); row blocks dealt statically chained heavy ones (61.7); drawn: 42.3, and
    // 37.8 with the partition cut at one tile of entries instead of two (tools/l12_trace.py, profiles/archive/r5s_*).
    a.blk = row_blocks;
    a.n_blk = n_row_blocks;
    l12_bwd_em_launch(a, F, Nf, row_blocks ? n_row_blocks : (n_rows + 255) / 256, st);
  } else {
    int64_t blocks = (n_rows * G + 255) / 256;
    if (blocks > kL12ResidentBlocks) blocks = kL12ResidentBlocks;      // all resident at 4 waves per SIMD; 1 280 - 2 048 blocks measured slower (32 - 36 us vs 31)
    l12_launch<true, false>(a, F, Nf, G, (unsigned)blocks, 0, st);
  }
  return check_launch("layer12_bwd");
}

#ifdef TMGCN_L12_TRACE
extern "C" int tmgcn_debug_l12_trace(unsigned long long* dst, long n_words, int clear) {   // clear & 2: the forward's stamps
  void* p = nullptr;
  hipError_t e = (clear & 2) ? hipGetSymbolAddress(&p, HIP_SYMBOL(tmgcn::l12_trace_fwd_words)) : hipGetSymbolAddress(&p, HIP_SYMBOL(tmgcn::l12_trace_words));
  if (e == hipSuccess && n_words > 0) e = hipMemcpy(dst, p, (size_t)n_words * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess && (clear & 1)) e = hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 16);
  return (int)e;
}
#endif

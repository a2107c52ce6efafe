// P2+P3 fused — Y = act((Â ⋆ X) · Wop), one launch, the SpMM result never leaves the CU
// on its way to the GEMM   (gfx950 / CDNA4)
//
// Replaces the pair  AtXt[k] = t.sparse.mm(At[k], Xt[k]) ; t.matmul(AtXt, W)
// (embedding_help_functions.py:206-207 + 222, 303-304 + 349, 471-472 + 486-489) and, fed the
// transposed CSR and Wᵀ, the backward pair: by associativity  Âᵀ(dY·Wᵀ) = (Âᵀ·dY)·Wᵀ,
// so the backward is the same kernel run on dY.
//
// Why fuse: the SpMM is an HBM-bound gather (MFMA idle), the f32 GEMM at F=128 is MFMA-bound
// (HBM idle); run back to back they cost the sum, fused the GEMM hides under the gather and
// the [T,N,F] intermediate is not re-read (it is still written once when the caller needs it
// for dW).  Structure per 64-row tile of a persistent 256-thread block:
//   phase 1  each wave gathers 16 rows — or, in a tile of few entries, the four waves' lane groups an equal share of the
//            tile's entries each (spmm_row.h) —, row sums go to an LDS tile [64][K+4]
//   phase 2  exact-f32 MFMA (v_mfma_f32_32x32x2_f32): wave w owns output columns
//            [32w, 32w+32); its B fragments (a 32-column strip of W) stay in registers for the
//            whole launch; A fragments come from the LDS tile with conflict-free ds_read_b128.
// The two phases of a block alternate, and on gfx950 an exact-f32 MFMA chain does not overlap another wave's vector work on the
// same SIMD (tools/probes/mfma_valu_overlap.hip): at S4's 33 entries per row the products are 5 % of a tile and disappear
// in the bandwidth-bound gather of the CU's other blocks; at the 4 entries per row of the reference's real operand gather and
// products add up (DESIGN.md §4 "Short tiles").
#include "common.h"
#include "spmm_row.h"

namespace tmgcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// tuning knobs (defaults = the measured best; tools/ab_variants.sh builds alternatives)
#ifndef TMGCN_FUSED_OCC
#define TMGCN_FUSED_OCC 4   // min waves per SIMD asked of the register allocator (A/B: 4 beats 3 by 4.5 %)
#endif
#ifndef TMGCN_FUSED_U
#define TMGCN_FUSED_U 4     // gathers in flight per lane (F = 64 / 128 variants)
#endif

constexpr int FBM = 64;         // rows per tile
constexpr int FKC = 128;        // max K (feature width of X)
constexpr int FLDA = FKC + 4;   // LDS row stride in floats

struct FusedArgs {
  const int64_t* rowptr;
  const int32_t* col;
  const float* val;
  const float4* X;
  int64_t n_rows;
  int32_t N;
  int32_t K;        // feature width of X (multiple of 4, <= 128)
  const float* W;
  int32_t Nf;       // output width (<= 128)
  int32_t trans_w;
  int64_t rows_per_batch;  // 0: one shared W; N: one W per slice
  int64_t w_batch_stride;
  float* Y;
  float* AX;        // optional: the SpMM result itself ([n_rows][K]), for dW
  float* pre;       // optional: pre-activation
  int32_t act;
  TileMap tiles;               // tiles restart at every slice (spmm_row.h), whatever the weight layout
  int64_t n_tiles;
  unsigned int* tile_counter;  // dynamic tile scheduling (common.h): two counters, [0] tiles, [1] scan windows (spmm_row.h)
  GiantPlan giant;             // rows summed chunk by chunk in front of this launch (spmm_row.h), or rows == nullptr
};

#ifndef TMGCN_FUSED_US
#define TMGCN_FUSED_US 1    // gathers in flight per lane on short tiles, as a multiple of U
#endif
#ifndef TMGCN_BX3_US
#define TMGCN_BX3_US 4           // gathers in flight per lane on its short tiles
#endif
#ifndef TMGCN_BX_STAGED_Y
#define TMGCN_BX_STAGED_Y 1      // the bf16-product kernel's Y tile leaves through LDS as whole rows (0: 64-byte pieces from the accumulators)
#endif
#ifndef TMGCN_BX_DRAW_AHEAD
#define TMGCN_BX_DRAW_AHEAD 1
#endif
#ifndef TMGCN_BX3_MAX_DEG
#define TMGCN_BX3_MAX_DEG 14     // launches with fewer entries per row (the caller's hint) take the bf16-product kernel (measured against the tile kernel: -9 % at 4, -6 % at 8, -4 % at 12; at 33 = S4 it would be 2 % faster too, but the adjoint identity at S4 size then holds to 4-7e-5 instead of 1e-5: the headline stays on the exact-f32 chain; profiles/r6/r6_44_*, not_kept/r6_73_*, not_kept/r6_74_*)
#endif
#ifndef TMGCN_FUSED_MFMA_PRIO
#define TMGCN_FUSED_MFMA_PRIO 3
#endif
#ifndef TMGCN_FUSED_BLOCKS
#define TMGCN_FUSED_BLOCKS 4   // development only: resident blocks per CU the grid is sized for (of 4)
#endif
#ifndef TMGCN_DEV_SKIP
#define TMGCN_DEV_SKIP 0    // development only (phase breakdown, profiles/r6/r6_05_*): 1 no gather, 2 no products, 4 no Y stores, 8 a third of the products
#endif

// Development build only (-DTMGCN_FUSED_TRACE, tools/fused_trace.py): thread 0 of every block sums the 100 MHz wall-clock time
// it spends in each phase of its tiles — separately for short tiles (entry-major walk) and the others — and leaves the
// sums in a device array read back through tmgcn_debug_fused_trace.  Not part of the library.
#ifdef TMGCN_FUSED_TRACE
__device__ unsigned long long fused_trace_words[4096 * 16];
#define FT_NOW() wall_clock64()
#define FT_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define FT_STAMP(name) const unsigned long long name = FT_NOW()
#else
#define FT_STAMP(name) do { } while (0)
#define FT_WAIT() do { } while (0)
#endif

// ---- the three pieces of the fused kernel ----------------------------------------------------------------------------

// W fragments of a wave's 32-column strip (n0 .. n0+31): B operand of v_mfma_f32_32x32x2_f32, k = 8j + s + 4·lh
template <int NJ>
__device__ __forceinline__ void fused_load_w(const FusedArgs& a, int64_t batch, int n0, int li, int lh, float (&wreg)[NJ][4]) {
  const float* Wb = a.W + (a.rows_per_batch ? batch * a.w_batch_stride : 0);
  const int n = n0 + li;
  const int nc = n < a.Nf ? n : 0;  // clamp: out-of-range columns load column 0, zeroed below
  const float* Wl = a.trans_w ? Wb + (int64_t)nc * a.K + 4 * lh : Wb + (int64_t)(4 * lh) * a.Nf + nc;
  const int64_t sk = a.trans_w ? 1 : a.Nf;  // stride of k
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float w = Wl[(int64_t)(8 * j + s) * sk];
      wreg[j][s] = n < a.Nf ? w : 0.f;
    }
}

// Phase 1: the row sums of one tile into the LDS tile `As` ([64][FLDA]) and, when asked for, to AX; all four waves.
// A tile of few entries entry-major, several rows per wave at once (spmm_row.h "Short tiles"); otherwise 16 rows per wave,
// long rows afterwards on all four waves.
template <int LPR, int U, int US>
__device__ __forceinline__ void fused_gather_tile(const FusedArgs& a, float* As, float4* s_part, const TileRows& rows, int64_t row0,
                                                  int64_t row_end, int lane, int wave, unsigned int* s_row) {
  const int F4 = a.K / 4;
  const int n_tile_rows = row_end - row0 < FBM ? (int)(row_end - row0) : FBM;
  const int64_t slice0 = row0 / a.N;
  const bool is_short = short_tile(rows, row0 + n_tile_rows <= (slice0 + 1) * a.N);
  if (TMGCN_DEV_SKIP & 1) return;
  if (is_short) {
    gather_short_tile<LPR, US>(a.col, a.val, a.X + slice0 * (int64_t)a.N * F4, rows, n_tile_rows, F4, lane, wave, F4,
                               [&](int rr, const float4& acc, int fl) {
                                 if (fl < F4) {
                                   *reinterpret_cast<float4*>(&As[rr * FLDA + 4 * fl]) = acc;
                                   if (a.AX) store_f4(&reinterpret_cast<float4*>(a.AX)[(row0 + rr) * F4 + fl], acc);
                                 }
                               });
    return;
  }
  // The rows of the tile are DRAWN by the four waves (an LDS counter, 4 at the start of a tile; the draw is issued in front of
  // the row it follows, so its latency hides under that row's gather): with a fixed deal — rows w, w + 4, … — the waves of a
  // skewed tile met 10.9 us apart at the barrier behind this loop (power-law graph, traced; equal rows: 1.5).  A row is summed
  // by whichever wave draws it, in the same order: the same bits.  Worth 0.3-0.5 % on the power-law graph (the launch is
  // bandwidth-bound: the CU's other blocks fill the wait), nothing on equal rows (profiles/r6/r6_35_*).
  for (int rr = wave; rr < FBM;) {
    unsigned int nxt = 0;
    if (lane == 0) nxt = atomicAdd(s_row, 1u);
    const int64_t r = row0 + rr;
    const bool lng = (rows.long_mask >> rr) & 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < row_end && !lng) {
      const int64_t slice = r / a.N;
      acc = gather_row<LPR, U>(a.col, a.val, a.X + slice * (int64_t)a.N * F4, readlane64(rows.beg, rr),
                               readlane64(rows.end, rr), F4, lane);
    }
    if (!lng && lane < LPR && lane < F4) {
      *reinterpret_cast<float4*>(&As[rr * FLDA + 4 * lane]) = acc;
      if (a.AX && r < row_end) store_f4(&reinterpret_cast<float4*>(a.AX)[r * F4 + lane], acc);
    }
    rr = (int)__builtin_amdgcn_readfirstlane(nxt);
  }
  for (uint64_t m = rows.long_mask; m; m &= m - 1) {
    const int rr = __builtin_ctzll(m);
    const int64_t r = row0 + rr;
    const int64_t slice = r / a.N;
    const int64_t beg = readlane64(rows.beg, rr), end = readlane64(rows.end, rr);
    float4 acc;
    const int gi = (a.giant.rows && end - beg > kGiantRow) ? giant_find(a.giant, r) : -1;   // uniform over the four waves
    if (gi >= 0) {
      if (wave != (rr & 3)) continue;
      acc = giant_row_sum(a.giant, gi, F4, lane, 0, F4);
    } else {
      acc = gather_long_row<LPR, U>(a.col, a.val, a.X + slice * (int64_t)a.N * F4, beg, end, F4, lane, wave, s_part);
    }
    if (wave == (rr & 3) && lane < LPR && lane < F4) {
      *reinterpret_cast<float4*>(&As[rr * FLDA + 4 * lane]) = acc;
      if (a.AX) store_f4(&reinterpret_cast<float4*>(a.AX)[r * F4 + lane], acc);
    }
  }
}

// The 16 accumulators of a lane after a 32-row half: accumulator i is row k(i) + 4·lh of the half, k(i) = (i & 3) + 8·(i >> 2).
template <bool GUARD, bool PRE, bool ACT>
__device__ __forceinline__ void fused_store_half(const f32x16& acc, const ActApply& act, float* __restrict__ Yb, float* __restrict__ Pb,
                                                 int Nf, int lane_off, int rows_left) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = (i & 3) + 8 * (i >> 2);
    const float s = acc[i];
    if ((TMGCN_DEV_SKIP & 4) && s != 12345.f) continue;
    if (GUARD && k >= rows_left) continue;
    if (PRE) store_f1(&Pb[k * Nf + lane_off], s);
    store_f1(&Yb[k * Nf + lane_off], ACT ? act(s) : s);
  }
}
template <bool GUARD>
__device__ __forceinline__ void fused_store_half(const f32x16& acc, int act_id, float* __restrict__ Yb, float* __restrict__ Pb, int Nf,
                                                 int lane_off, int rows_left) {
  const ActApply act(act_id);             // decoded once (the same bits as act_apply); no activation: the raw sums, no select chain
  if (act_id == TMGCN_ACT_NONE) {
    if (Pb) fused_store_half<GUARD, true, false>(acc, act, Yb, Pb, Nf, lane_off, rows_left);
    else fused_store_half<GUARD, false, false>(acc, act, Yb, Pb, Nf, lane_off, rows_left);
  } else {
    if (Pb) fused_store_half<GUARD, true, true>(acc, act, Yb, Pb, Nf, lane_off, rows_left);
    else fused_store_half<GUARD, false, true>(acc, act, Yb, Pb, Nf, lane_off, rows_left);
  }
}

// Phase 2: tile · Wop on the matrix cores, the wave's 32 output columns [n0, n0 + 32).
// One 32-row half of the tile at a time: its 16 accumulators are stored before the other half's products
// start, so only ONE accumulator set is live next to the 64 W-fragment registers (both halves live — the
// round 1-3 form — cost 15 spilled VGPRs at 4 waves per SIMD; profiles/archive/r4*_ab_fused_spill.txt).
// A fragments are fetched one k-group ahead of the MFMAs that use them; the sched_barrier keeps hipcc
// from hoisting all the ds_read_b128 to the top.
template <int NJ>
__device__ __forceinline__ void fused_mfma_tile(const FusedArgs& a, const float* As, const float (&wreg)[NJ][4], int64_t row0,
                                                int64_t row_end, int n0, int li, int lh) {
  if (n0 >= a.Nf || (TMGCN_DEV_SKIP & 2)) return;
  const float* Arow = &As[li * FLDA + 4 * lh];
  const int n = n0 + li;
#pragma unroll
  for (int mb = 0; mb < FBM / 32; ++mb) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 av_next = *reinterpret_cast<const float4*>(Arow + mb * 32 * FLDA);
#pragma unroll
    for (int j = 0; j < ((TMGCN_DEV_SKIP & 8) ? (NJ + 2) / 3 : NJ); ++j) {      // (8: a THIRD of the products — a timing bound, wrong results)
      const float4 av = av_next;
      if (j + 1 < NJ) av_next = *reinterpret_cast<const float4*>(Arow + mb * 32 * FLDA + 8 * (j + 1));
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wreg[j][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wreg[j][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wreg[j][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wreg[j][3], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (n < a.Nf) {
      // Epilogue: accumulator i of the lane is row rbase + k(i) + 4·lh, column n.  Everything but (4·lh)·Nf + n is uniform: the
      // stores take a scalar base + a 32-bit lane offset, the activation is decoded once (ActApply: the same bits as
      // act_apply; none at all for TMGCN_ACT_NONE), and a half tile that lies inside the slice skips the row guard.  (Round 6:
      // a 64-bit address, a row compare and an activation switch per ELEMENT had made the epilogues a third of the product
      // phase: 6.7 us per tile on an otherwise idle CU where the MFMAs need 3.6.)
      const int64_t rbase = row0 + mb * 32;
      float* __restrict__ Yb = a.Y + rbase * a.Nf;
      float* __restrict__ Pb = a.pre ? a.pre + rbase * a.Nf : nullptr;
      const int lane_off = (4 * lh) * a.Nf + n;
      if (rbase + 32 <= row_end) fused_store_half<false>(acc, a.act, Yb, Pb, a.Nf, lane_off, 32);      // uniform: inside the slice
      else fused_store_half<true>(acc, a.act, Yb, Pb, a.Nf, lane_off, (int)(row_end - rbase) - 4 * lh);  // rows k < rows_left exist
    }
  }
}

template <int LPR, int U, int NJ, int US = TMGCN_FUSED_US * U>  // NJ = K / 8 (K is a multiple of 8 here)
__global__ __launch_bounds__(256, TMGCN_FUSED_OCC) void spmm_gemm_kernel(FusedArgs a) {
  __shared__ float As[FBM * FLDA];
  __shared__ float4 s_part[4 * LPR];      // partial sums of a long row, one per wave (spmm_row.h)
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = wave * 32;
  const TileMap tm = a.tiles;

  float wreg[NJ][4];
  int64_t cur_batch = -1;
  __shared__ unsigned int s_tile, s_row;
  if (threadIdx.x == 0) s_row = 4;        // (the first use is behind the barrier of the first tile draw)
  HeavyScan heavy;
  heavy.init(a.rowptr, tm);

#ifdef TMGCN_FUSED_TRACE
  unsigned long long ft[16] = {0};        // [8·short + phase]: 0 draw, 1 row pointers, 2 gather (wave 0), 3 barrier, 4 products, 5 barrier, 6 tiles
#endif
  for (;;) {
    // The lane index is laundered through an empty asm once per tile and once more in front of the product phase: what the
    // two phases derive from it (feature lane, stream, LDS and output addresses) is then recomputed where it is used — a few
    // VALU instructions — instead of being hoisted out of this loop and held in registers across the OTHER phase, where the
    // 64 W fragments and the gather's loads in flight need them (round 6: two W fragments lived in scratch and were
    // re-read inside the MFMA chain, four exposed loads per tile).
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    FT_STAMP(ft_a);
    // next tile: first the heavy tiles (spmm_row.h: windows drawn from counter[1]), then from the device counter
    // (counter[0]; ascending, so resident blocks stay inside one slice)
    int64_t tile = -1;
    if (heavy.scanning) tile = heavy.next(a.rowptr, tm, a.tile_counter + 1, &s_tile, lane);
    const bool scanning = heavy.scanning;
    if (!scanning) {
      if (threadIdx.x == 0) s_tile = atomicAdd(a.tile_counter, 1u);
      __syncthreads();
      tile = s_tile;
      if (tile >= a.n_tiles) break;
    }
    int64_t unit, row0, row_end;
    tile_extent(tm, tile, unit, row0, row_end);
    const int64_t batch = a.rows_per_batch ? row0 / a.rows_per_batch : 0;
    FT_STAMP(ft_b);
    TileRows rows;
    rows.load(a.rowptr, row0, row_end, lane);
    if (TMGCN_HEAVY_FIRST && !scanning && rows.entries > heavy.thr) {   // done in somebody's pass 1
      __syncthreads();                                                 // (s_tile is rewritten at the top)
      continue;
    }
    FT_WAIT();
    FT_STAMP(ft_c);
    if (batch != cur_batch) {
      fused_load_w<NJ>(a, batch, n0, lane & 31, lane >> 5, wreg);
      cur_batch = batch;
    }
    fused_gather_tile<LPR, U, US>(a, As, s_part, rows, row0, row_end, lane, wave, &s_row);
    FT_WAIT();
    FT_STAMP(ft_d);
    __syncthreads();
    if (threadIdx.x == 0) s_row = 4;        // for the next tile's row draws (two barriers away)
    FT_STAMP(ft_e);
    // the product phase at raised issue priority: its waves hold the block's LDS tile and share the SIMD with three other
    // blocks' waves that are waiting for gathered rows anyway (round 6: -4.5 % on the chess operand at bench size, -6 % at
    // 4 random entries per row, S4 unchanged; profiles/r6/r6_08_*)
    __builtin_amdgcn_s_setprio(TMGCN_FUSED_MFMA_PRIO);
    {
      int lane_p = lane0;
      asm volatile("" : "+v"(lane_p));
      fused_mfma_tile<NJ>(a, As, wreg, row0, row_end, n0, lane_p & 31, lane_p >> 5);
    }
    __builtin_amdgcn_s_setprio(0);
    FT_STAMP(ft_f);
    __syncthreads();  // tile consumed before the next phase 1 overwrites it
#ifdef TMGCN_FUSED_TRACE
    {
      const int n_tile_rows = row_end - row0 < FBM ? (int)(row_end - row0) : FBM;
      const int k = short_tile(rows, row0 + n_tile_rows <= (row0 / a.N + 1) * a.N) ? 8 : 0;
      const unsigned long long ft_g = FT_NOW();
      ft[k + 0] += ft_b - ft_a;
      ft[k + 1] += ft_c - ft_b;
      ft[k + 2] += ft_d - ft_c;
      ft[k + 3] += ft_e - ft_d;
      ft[k + 4] += ft_f - ft_e;
      ft[k + 5] += ft_g - ft_f;
      ft[k + 6] += 1;
    }
#endif
  }
#ifdef TMGCN_FUSED_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 4096)
    for (int i = 0; i < 16; ++i) fused_trace_words[blockIdx.x * 16 + i] = ft[i];
#endif
}


// ---- the low-degree kernel (round 6): the products on the bf16 matrix cores ---------------------------------------------------
// On gfx950 an exact-f32 MFMA chain and another wave's vector work on the same SIMD serialize (tools/probes/mfma_valu_overlap.hip),
// so at few entries per row — the reference's real operand has 4 — the tile kernel's launch is its gather PLUS its products
// (8 + 9 ms on that operand at bench size).  This kernel does the products the way the library's standalone GEMM does
// (gemm.hip gemm_bf16x3): every fp32 operand split exactly into three bf16 planes, the six plane products that matter on
// v_mfma_f32_16x16x32_bf16 — a third of the matrix-pipe time, on a pipe of its own — i.e. the numerics of the unfused
// default route (SpMM + bf16-split GEMM: within 1e-5 of the oracle, measured 3e-7), not the exact-f32 chain of the tile kernel.
// Shape: 512-thread blocks, two per CU.  All eight waves gather (the same row walks in the same order: AX stays bit-equal to
// the plain kernel's); a row's sum is split when it is flushed and written as three bf16 plane images [64][272 B] (gemm.hip's
// conflict-free pitch); then each wave multiplies the tile by ITS 16-column strip of Wop, whose planes it holds in 12·K/32
// registers (48 at K = 128 — fewer than the 64 f32 fragments of the tile kernel, which is what lets eight waves fit).
// The Y tile does not leave from the accumulators (a wave owns 16 columns: 64-byte pieces of every row) but through LDS, the planes'
// memory, as whole rows — see the epilogue; the main loop's next tile is drawn while the products run.
// Measured and not kept for this regime: a 16-wave ring kernel, gather waves and product waves decoupled through 16-row LDS slots
// (profiles/r6/not_kept/r6_66_low_degree_ring_kernel_not_kept.patch): bit-equal, never faster — the gather runs at the plain SpMM's rate
// either way, the rest is products, barriers and stores.
typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bx_f32x2 __attribute__((ext_vector_type(2)));
typedef float bx_f32x4 __attribute__((ext_vector_type(4)));
constexpr int BX_PITCH = 272;                  // bytes per row of a plane image: 128 bf16 + 16 B pad
constexpr int BX_PLANE = FBM * BX_PITCH;       // 17 408 B
constexpr int BX_YPITCH = 132;                 // floats per row of the Y tile handed over through LDS (33 792 B of the planes' 52 224): rows 4 apart are 16 banks apart

__device__ __forceinline__ unsigned bx_pack(float a, float b) {  // bf16(a) | bf16(b) << 16, RNE
  bx_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx_bf16x2));
}
__device__ __forceinline__ void bx_split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {   // x = hi + mid + lo, exactly
  h = bx_pack(a, b);
  a -= __uint_as_float(h << 16);
  b -= __uint_as_float(h & 0xffff0000u);
  m = bx_pack(a, b);
  a -= __uint_as_float(m << 16);
  b -= __uint_as_float(m & 0xffff0000u);
  l = bx_pack(a, b);
}

template <int LPR, int U, int NKS, int US>     // NKS = K / 32
__global__ __launch_bounds__(512, 4) void spmm_gemm_bx3_kernel(FusedArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[3 * BX_PLANE];
  __shared__ float4 s_part[4 * LPR];
  __shared__ unsigned int s_tile, s_row;
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // 0..7
  const int n0 = wave * 16;                                                // this wave's column strip
  const int F4 = a.K / 4;
  const TileMap tm = a.tiles;
  unsigned bw[NKS][3][4];                                                  // B fragments of Wop: [k-step][plane][8 bf16]
  int64_t cur_batch = -1;
  if (threadIdx.x == 0) s_row = 8;
  // whole rows of aligned float4: Y leaves through LDS (see the epilogue)
  const bool staged = TMGCN_BX_STAGED_Y && a.Nf % 4 == 0 && reinterpret_cast<uintptr_t>(a.Y) % 16 == 0 && reinterpret_cast<uintptr_t>(a.pre) % 16 == 0;
  HeavyScan heavy;                                                         // the heaviest tiles first, as in the tile kernel
  heavy.init(a.rowptr, tm);
  bool have_next = false;                                                  // s_tile holds the main loop's next tile
  for (;;) {
    int lane = lane0;                                                      // laundered per tile (see the tile kernel)
    asm volatile("" : "+v"(lane));
    int64_t tile = -1;
    if (heavy.scanning) tile = heavy.next(a.rowptr, tm, a.tile_counter + 1, &s_tile, lane);
    const bool scanning = heavy.scanning;
    if (!scanning) {
      if (!have_next) {                                                    // (else: drawn under the products of the tile before)
        if (threadIdx.x == 0) s_tile = atomicAdd(a.tile_counter, 1u);
        __syncthreads();
      }
      tile = s_tile;
      have_next = false;
      if (tile >= a.n_tiles) break;
    }
    int64_t unit, row0, row_end;
    tile_extent(tm, tile, unit, row0, row_end);
    const int64_t batch = a.rows_per_batch ? row0 / a.rows_per_batch : 0;
    TileRows rows;
    rows.load(a.rowptr, row0, row_end, lane);
    if (TMGCN_HEAVY_FIRST && !scanning && rows.entries > heavy.thr) {     // done in somebody's pass 1
      __syncthreads();
      continue;
    }
    const int n_tile_rows = row_end - row0 < FBM ? (int)(row_end - row0) : FBM;
    const int64_t slice0 = row0 / a.N;
    const float4* Xs = a.X + slice0 * (int64_t)a.N * F4;                   // (tiles restart at every slice)
    // a row's sum: three plane images in LDS (8 bytes per plane and lane) and, when asked for, AX
    auto flush = [&](int rr, const float4& acc, int fl) __attribute__((always_inline)) {
      if (fl < F4) {
        unsigned h0, m0, l0, h1, m1, l1;
        bx_split3(acc.x, acc.y, h0, m0, l0);
        bx_split3(acc.z, acc.w, h1, m1, l1);
        unsigned char* w = sm + rr * BX_PITCH + fl * 8;
        *reinterpret_cast<uint2*>(w) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(w + BX_PLANE) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(w + 2 * BX_PLANE) = make_uint2(l0, l1);
        if (a.AX && rr < n_tile_rows) store_f4(&reinterpret_cast<float4*>(a.AX)[(row0 + rr) * F4 + fl], acc);
      }
    };
    if (short_tile(rows, row0 + n_tile_rows <= (slice0 + 1) * a.N)) {
      gather_short_tile<LPR, US, 8>(a.col, a.val, Xs, rows, n_tile_rows, F4, lane, wave, F4, flush);
    } else {
      for (int rr = wave; rr < FBM;) {                                    // rows drawn by the eight waves (LDS counter)
        unsigned int nxt = 0;
        if (lane == 0) nxt = atomicAdd(&s_row, 1u);
        const bool lng = (rows.long_mask >> rr) & 1;
        if (!lng) {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rr < n_tile_rows) acc = gather_row<LPR, U>(a.col, a.val, Xs, readlane64(rows.beg, rr), readlane64(rows.end, rr), F4, lane);
          if (lane < LPR) flush(rr, acc, lane);
        }
        rr = (int)__builtin_amdgcn_readfirstlane(nxt);
      }
      for (uint64_t m = rows.long_mask; m; m &= m - 1) {                  // long rows: waves 0-3 share each (the same quarters, the
        const int rr = __builtin_ctzll(m);                                // same order as everywhere); waves 4-7 keep the barriers company
        const int64_t beg = readlane64(rows.beg, rr), end = readlane64(rows.end, rr);
        if (wave < 4) {
          const float4 acc = gather_long_row<LPR, U>(a.col, a.val, Xs, beg, end, F4, lane, wave, s_part);
          if (wave == (rr & 3) && lane < LPR) flush(rr, acc, lane);
        } else {
          __syncthreads();
          __syncthreads();
        }
      }
    }
    if (batch != cur_batch) {                                              // Wop[k][n0 + n]: this lane's column, k = 32 ks + 8 (lane / 16) + 2 j, + 1
      const float* Wb = a.W + (a.rows_per_batch ? batch * a.w_batch_stride : 0);
      const int n = n0 + (lane & 15);
      const int nc = n < a.Nf ? n : 0;
      const float zn = n < a.Nf ? 1.f : 0.f;
      const int64_t sk = a.trans_w ? 1 : a.Nf, sn = a.trans_w ? a.K : 1;
      const float* Wl = Wb + (int64_t)nc * sn + (int64_t)(8 * (lane >> 4)) * sk;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float w0 = Wl[(int64_t)(32 * ks + 2 * j) * sk] * zn;
          const float w1 = Wl[(int64_t)(32 * ks + 2 * j + 1) * sk] * zn;
          bx_split3(w0, w1, bw[ks][0][j], bw[ks][1][j], bw[ks][2][j]);
        }
      cur_batch = batch;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_row = 8;
    // the main loop's next tile is drawn now, under the products (one barrier and the draw's round trip less per tile)
    unsigned int drawn = 0;
    if (TMGCN_BX_DRAW_AHEAD && !scanning && threadIdx.x == 0) drawn = atomicAdd(a.tile_counter, 1u);
    // ---- products: 4 row blocks of 16 x this wave's 16 columns; per (row block, k-step) six plane products, small terms first
    __builtin_amdgcn_s_setprio(TMGCN_FUSED_MFMA_PRIO);
    int lane_p = lane0;
    asm volatile("" : "+v"(lane_p));
    const int lm = lane_p & 15, lg = lane_p >> 4;
    bx_f32x4 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = bx_f32x4{0.f, 0.f, 0.f, 0.f};
    if (n0 < a.Nf) {
      const unsigned char* rd = sm + lm * BX_PITCH + lg * 16;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bx_bf16x8 bh = __builtin_bit_cast(bx_bf16x8, make_uint4(bw[ks][0][0], bw[ks][0][1], bw[ks][0][2], bw[ks][0][3]));
        const bx_bf16x8 bm = __builtin_bit_cast(bx_bf16x8, make_uint4(bw[ks][1][0], bw[ks][1][1], bw[ks][1][2], bw[ks][1][3]));
        const bx_bf16x8 bl = __builtin_bit_cast(bx_bf16x8, make_uint4(bw[ks][2][0], bw[ks][2][1], bw[ks][2][2], bw[ks][2][3]));
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const unsigned char* p = rd + rb * 16 * BX_PITCH + ks * 64;
          const bx_bf16x8 ah = __builtin_bit_cast(bx_bf16x8, *reinterpret_cast<const uint4*>(p));
          const bx_bf16x8 am = __builtin_bit_cast(bx_bf16x8, *reinterpret_cast<const uint4*>(p + BX_PLANE));
          const bx_bf16x8 al = __builtin_bit_cast(bx_bf16x8, *reinterpret_cast<const uint4*>(p + 2 * BX_PLANE));
          bx_f32x4 c = acc[rb];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
          acc[rb] = c;
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // epilogue: accumulator i of block rb is row 16 rb + 4 lg + i, column n0 + lm
    if (TMGCN_BX_STAGED_Y == 2 && staged) {
      // development (measured: the same time as the dword stores, profiles/r6/r6_82_*): no LDS, no barrier — the 4 x 4 blocks
      // transposed on the lane quads (common.h), 16 bytes per lane: a quarter of the store instructions, every row still leaving as
      // 64-byte pieces.  It is the pieces that cost, not the instruction count.
      const int j = lm & 3, cq = n0 + 4 * (lm >> 2);
      if (cq < a.Nf) {
        const ActApply act(a.act);
        float* __restrict__ Yb = a.Y + row0 * a.Nf;
        float* __restrict__ Pb = a.pre ? a.pre + row0 * a.Nf : nullptr;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          float v[4] = {acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]};
          quad_transpose4(v, j);
          const int rr = 16 * rb + 4 * lg + j;
          if (rr < n_tile_rows) {
            float4 q = make_float4(v[0], v[1], v[2], v[3]);
            if (Pb) store_f4(reinterpret_cast<float4*>(&Pb[rr * a.Nf + cq]), q);
            if (a.act != TMGCN_ACT_NONE) q = make_float4(act(q.x), act(q.y), act(q.z), act(q.w));
            store_f4(reinterpret_cast<float4*>(&Yb[rr * a.Nf + cq]), q);
          }
        }
      }
    } else if (!staged) {                                                // (an output that is not a whole number of aligned float4 per row)
      const int n = n0 + lm;
      if (n < a.Nf) {
        const ActApply act(a.act);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rr = 16 * rb + 4 * lg + i;
            if (rr < n_tile_rows) {
              const float v = acc[rb][i];
              const int64_t o = (row0 + rr) * a.Nf + n;
              if (a.pre) store_f1(&a.pre[o], v);
              store_f1(&a.Y[o], a.act == TMGCN_ACT_NONE ? v : act(v));
            }
          }
      }
    } else {
      // A wave owns 16 columns: stored from the accumulators, every row of Y is written as eight 64-byte pieces, a dword per lane
      // (the tile kernel's 32-column strips write whole 128-byte lines).  Handed over through LDS — the planes' memory, free once
      // every wave has read it — the tile is stored row by row, 16 bytes per lane: whole lines, a quarter of the store
      // instructions.  Two more barriers per tile, and still 6-7 % off the launch (chess operand 13.76 -> 12.92 ms; profiles/r6/r6_70_*, r6_82_*).
      __syncthreads();                                                   // every wave is through with the planes
      float* yt = reinterpret_cast<float*>(sm);
      if (n0 < a.Nf) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int i = 0; i < 4; ++i) yt[(16 * rb + 4 * lg + i) * BX_YPITCH + n0 + lm] = acc[rb][i];
      }
      __syncthreads();                                                   // the tile is in LDS
      int lane_s = lane0;
      asm volatile("" : "+v"(lane_s));
      const int c4 = lane_s & 31;                                        // float4 of the row; two rows per wave and pass
      if (4 * c4 < a.Nf) {
        const ActApply act(a.act);
        float* __restrict__ Yb = a.Y + row0 * a.Nf;
        float* __restrict__ Pb = a.pre ? a.pre + row0 * a.Nf : nullptr;
#pragma unroll
        for (int k = 0; k < FBM / 16; ++k) {
          const int rr = 16 * k + 2 * wave + (lane_s >> 5);
          if (rr < n_tile_rows) {
            float4 v = *reinterpret_cast<const float4*>(&yt[rr * BX_YPITCH + 4 * c4]);
            if (TMGCN_DEV_SKIP & 4) continue;
            if (Pb) store_f4(reinterpret_cast<float4*>(&Pb[rr * a.Nf + 4 * c4]), v);
            if (a.act != TMGCN_ACT_NONE) v = make_float4(act(v.x), act(v.y), act(v.z), act(v.w));
            store_f4(reinterpret_cast<float4*>(&Yb[rr * a.Nf + 4 * c4]), v);
          }
        }
      }
    }
    if (TMGCN_BX_DRAW_AHEAD && !scanning) {
      if (threadIdx.x == 0) s_tile = drawn;                                // (everybody read the current tile's number before the barrier behind the gather)
      have_next = true;
    }
    __syncthreads();      // the planes (or the Y tile in their place) are consumed before the next tile's rows overwrite them
  }
}

// ---------------------------------------------------------------------------------------------
// Small-F variant (the reference's real widths: F = 2 -> 6 -> 6): G lanes share a
// row and stride over its non-zeros as in spmm_small; after the shuffle butterfly every lane of
// the group holds the row sum, and lane gl produces output columns gl, gl+G, ...  The weight is
// a few dozen floats, read through L1.  One launch instead of SpMM + GEMM, no [T,N,F] round trip.
// ---------------------------------------------------------------------------------------------
template <int F, int G>
__global__ __launch_bounds__(256) void spmm_gemm_small_kernel(FusedArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = gid / G;
  const int gl = (int)(gid % G);
  const bool live = r < a.n_rows;
  const float* X = reinterpret_cast<const float*>(a.X);
  const ActApply act(a.act);    // decoded once: no switch per element (common.h)
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  int64_t beg = 0, end = 0;
  if (live) {
    beg = a.rowptr[r];
    end = a.rowptr[r + 1];
  }
  // rows that would take this group more than kNarrowLong trips are left to the whole wave below (spmm_row.h)
  const bool is_long = G < kWave && end - beg > (int64_t)kNarrowLong * G;
  if (live && !is_long) {
    const int64_t xoff = (r / a.N) * (int64_t)a.N;
    for (int64_t p = beg + gl; p < end; p += G) {
      const float* x = X + (xoff + a.col[p]) * F;
      const float v = a.val[p];
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] = fmaf(v, x[f], acc[f]);
    }
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] += __shfl_xor(acc[f], o);
  // the row sum is in every lane of the group (of the wave, for a long row): `lanes` of them form the output columns
  auto epilogue = [&](int64_t row, int first, int lanes) {
    if (a.AX && first == 0) {
#pragma unroll
      for (int f = 0; f < F; ++f) a.AX[row * F + f] = acc[f];
    }
    const float* Wb = a.W + (a.rows_per_batch ? (row / a.rows_per_batch) * a.w_batch_stride : 0);
    for (int n = first; n < a.Nf; n += lanes) {
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < F; ++f) s = fmaf(acc[f], a.trans_w ? Wb[(int64_t)n * F + f] : Wb[(int64_t)f * a.Nf + n], s);
      if (a.pre) a.pre[row * a.Nf + n] = s;
      a.Y[row * a.Nf + n] = act(s);
    }
  };
  if (live && !is_long) epilogue(r, gl, G);
  if constexpr (G < kWave) {
    const int lane = threadIdx.x & 63;
    for (uint64_t m = __ballot(is_long && gl == 0); m; m &= m - 1) {
      const int src = __builtin_ctzll(m);
      const int64_t r2 = readlane64(r, src), b2 = readlane64(beg, src), e2 = readlane64(end, src);
      narrow_wave_row<F>(acc, a.col, a.val, X, (r2 / a.N) * (int64_t)a.N, b2, e2, lane);
      epilogue(r2, lane, kWave);
    }
  }
}

template <int F>
static int launch_fused_small(const FusedArgs& a, int G, hipStream_t st) {
  const unsigned grid = (unsigned)((a.n_rows * G + 255) / 256);
  switch (G) {
    case 1: hipLaunchKernelGGL((spmm_gemm_small_kernel<F, 1>), dim3(grid), dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((spmm_gemm_small_kernel<F, 2>), dim3(grid), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((spmm_gemm_small_kernel<F, 4>), dim3(grid), dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((spmm_gemm_small_kernel<F, 8>), dim3(grid), dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL((spmm_gemm_small_kernel<F, 16>), dim3(grid), dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL((spmm_gemm_small_kernel<F, 32>), dim3(grid), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((spmm_gemm_small_kernel<F, 64>), dim3(grid), dim3(256), 0, st, a);
  }
  return check_launch("spmm_gemm_small");
}

static bool fused_small_ok(int K, int Nf) {
  return (K == 1 || K == 2 || K == 3 || K == 4 || K == 6 || K == 8) && Nf >= 1 && Nf <= 16;
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_spmm_gemm_supported(int32_t K, int32_t Nf) {
  return ((K % 8 == 0 && K >= 16 && K <= FKC && Nf >= 1 && Nf <= 128) || fused_small_ok(K, Nf)) ? 1 : 0;
}

extern "C" int tmgcn_spmm_gemm_f32_plan(const int64_t* rowptr, const int32_t* col, const float* val,
                                         const float* X, int64_t n_rows, int32_t N, int32_t K,
                                         const float* W, int32_t Nf, int32_t trans_w,
                                         int64_t rows_per_batch, int64_t w_batch_stride, int32_t act,
                                         float* Y, float* AX, float* pre_act, int32_t grid_reserve,
                                         float avg_nnz_per_row, const int64_t* giant_rows, const int32_t* giant_chunks,
                                         int32_t n_giant, int32_t n_giant_chunks, float* giant_ws, int64_t giant_ws_bytes,
                                         void* stream) {
  TMGCN_REQUIRE(grid_reserve >= 0 && grid_reserve <= 4096, "spmm_gemm: grid_reserve %d out of range [0, 4096]", grid_reserve);
  TMGCN_REQUIRE(n_rows >= 0 && N > 0, "spmm_gemm: bad shape n_rows=%lld N=%d", (long long)n_rows, N);
  TMGCN_REQUIRE(tmgcn_spmm_gemm_supported(K, Nf),
                "spmm_gemm: unsupported widths K=%d Nf=%d (need K a multiple of 8 in [16,128] with Nf <= 128, or K in {1,2,3,4,6,8} with Nf <= 16); "
                "use tmgcn_spmm_csr_batched_f32 + tmgcn_gemm_f32", K, Nf);
  TMGCN_REQUIRE(act >= TMGCN_ACT_NONE && act <= TMGCN_ACT_SELU, "spmm_gemm: unknown activation %d", act);
  TMGCN_REQUIRE(rows_per_batch >= 0, "spmm_gemm: negative rows_per_batch");
  if (n_rows == 0) return TMGCN_OK;
  TMGCN_REQUIRE(rowptr && X && W && Y, "spmm_gemm: null pointer");
  TMGCN_REQUIRE(n_rows % N == 0, "spmm_gemm: n_rows=%lld is not a multiple of N=%d", (long long)n_rows, N);
  if (fused_small_ok(K, Nf)) {
    FusedArgs s{rowptr, col, val, reinterpret_cast<const float4*>(X), n_rows, N, K, W, Nf, trans_w,
                rows_per_batch, w_batch_stride, Y, AX, pre_act, act, TileMap{0, 0, 0, 0}, 0, nullptr, GiantPlan{nullptr, nullptr, nullptr, 0}};
    // lanes per row: from the caller's average row length where it is known (as in
    // tmgcn_spmm_csr_batched_f32_hint), else 8 — the row lengths live on the device and the
    // reference's M-transformed adjacencies have tens of entries per row
    int G = 8;
    if (avg_nnz_per_row >= 0.f) {
      G = 1;
      while (G < 64 && (float)(2 * G) <= avg_nnz_per_row) G <<= 1;
    }
    hipStream_t st2 = (hipStream_t)stream;
    switch (K) {
      case 1: return launch_fused_small<1>(s, G, st2);
      case 2: return launch_fused_small<2>(s, G, st2);
      case 3: return launch_fused_small<3>(s, G, st2);
      case 4: return launch_fused_small<4>(s, G, st2);
      case 6: return launch_fused_small<6>(s, G, st2);
      default: return launch_fused_small<8>(s, G, st2);
    }
  }
  TMGCN_REQUIRE(reinterpret_cast<uintptr_t>(X) % 16 == 0 && (!AX || reinterpret_cast<uintptr_t>(AX) % 16 == 0),
                "spmm_gemm: X / AX must be 16-byte aligned");
  FusedArgs a{rowptr, col, val, reinterpret_cast<const float4*>(X), n_rows, N, K, W, Nf, trans_w,
              rows_per_batch, w_batch_stride, Y, AX, pre_act, act, TileMap{0, 0, 0, 0}, 0, nullptr, GiantPlan{nullptr, nullptr, nullptr, 0}};
  if (n_giant > 0) {
    const int rc = launch_giant_partial("spmm_gemm (giant rows)", rowptr, col, val, X, N, K, giant_rows, giant_chunks, n_giant,
                                        n_giant_chunks, giant_ws, giant_ws_bytes, (hipStream_t)stream);
    if (rc != TMGCN_OK) return rc;
    a.giant = GiantPlan{giant_rows, giant_chunks, reinterpret_cast<const float4*>(giant_ws), n_giant};
  }
  // a unit of tiles = a slice, unless the caller's weight batches do not end on slice boundaries (no layer does that)
  a.tiles = make_tile_map(n_rows, (rows_per_batch == 0 || rows_per_batch % N == 0) ? (int64_t)N : rows_per_batch);
  a.n_tiles = a.tiles.n_tiles;
  TMGCN_REQUIRE(a.n_tiles < (int64_t)0x7fffffff, "spmm_gemm: too many row tiles");
  a.tile_counter = acquire_tile_counters((hipStream_t)stream, 2);      // [0] the main loop's tiles, [1] the heavy-tile scan windows
  TMGCN_REQUIRE(a.tile_counter, "spmm_gemm: no tile counter: %s", pool_error());
  hipStream_t st = (hipStream_t)stream;
  // few entries per row (the caller's hint), no giant-row plan, K = 64 or 128: the products on the bf16 matrix cores
  if (avg_nnz_per_row >= 0.f && avg_nnz_per_row < (float)TMGCN_BX3_MAX_DEG && n_giant == 0 && (K == 128 || K == 64)) {
    if (K == 128) {
      int64_t gx = persistent_grid(spmm_gemm_bx3_kernel<32, TMGCN_FUSED_U, 4, TMGCN_BX3_US>, 512, 0, 2) - grid_reserve / 2;   // (a block of this kernel fills two of the tile kernel's slots)
      if (gx < 64) gx = 64;
      if (gx > a.n_tiles) gx = a.n_tiles;
      hipLaunchKernelGGL((spmm_gemm_bx3_kernel<32, TMGCN_FUSED_U, 4, TMGCN_BX3_US>), dim3((unsigned)gx), dim3(512), 0, st, a);
    } else {
      int64_t gx = persistent_grid(spmm_gemm_bx3_kernel<16, TMGCN_FUSED_U, 2, TMGCN_BX3_US>, 512, 0, 2) - grid_reserve / 2;
      if (gx < 64) gx = 64;
      if (gx > a.n_tiles) gx = a.n_tiles;
      hipLaunchKernelGGL((spmm_gemm_bx3_kernel<16, TMGCN_FUSED_U, 2, TMGCN_BX3_US>), dim3((unsigned)gx), dim3(512), 0, st, a);
    }
    return check_launch("spmm_gemm (bf16-split products)");
  }
  // persistent blocks: up to 4 per CU (LDS 33.8 KB each); tiles are drawn in ascending order so the
  // blocks resident at any moment work on neighbouring rows of the same slice
#define TMGCN_FUSED_CASE(KK, L, UU)                                                              \
  case KK: {                                                                                     \
    int64_t gx = persistent_grid_reserved(spmm_gemm_kernel<L, UU, KK / 8>, 256, grid_reserve);                 \
    gx = gx * TMGCN_FUSED_BLOCKS / 4;                                                            \
    if (gx > a.n_tiles) gx = a.n_tiles;                                                          \
    hipLaunchKernelGGL((spmm_gemm_kernel<L, UU, KK / 8>), dim3((unsigned)gx), dim3(256), 0, st, a); \
    break;                                                                                       \
  }
  switch (K) {  // every multiple of 8 in [16, 128]; lanes per feature row = next power of two >= K/4
    TMGCN_FUSED_CASE(16, 4, 2)
    TMGCN_FUSED_CASE(24, 8, 2)
    TMGCN_FUSED_CASE(32, 8, 2)
    TMGCN_FUSED_CASE(40, 16, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(48, 16, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(56, 16, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(64, 16, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(72, 32, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(80, 32, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(88, 32, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(96, 32, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(104, 32, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(112, 32, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(120, 32, TMGCN_FUSED_U)
    TMGCN_FUSED_CASE(128, 32, TMGCN_FUSED_U)
  }
#undef TMGCN_FUSED_CASE
  return check_launch("spmm_gemm");
}

extern "C" int tmgcn_spmm_gemm_f32_hint(const int64_t* rowptr, const int32_t* col, const float* val,
                                         const float* X, int64_t n_rows, int32_t N, int32_t K,
                                         const float* W, int32_t Nf, int32_t trans_w,
                                         int64_t rows_per_batch, int64_t w_batch_stride, int32_t act,
                                         float* Y, float* AX, float* pre_act, int32_t grid_reserve,
                                         float avg_nnz_per_row, void* stream) {
  return tmgcn_spmm_gemm_f32_plan(rowptr, col, val, X, n_rows, N, K, W, Nf, trans_w, rows_per_batch, w_batch_stride, act, Y, AX,
                                  pre_act, grid_reserve, avg_nnz_per_row, nullptr, nullptr, 0, 0, nullptr, 0, stream);
}

extern "C" int tmgcn_spmm_gemm_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                                    const float* X, int64_t n_rows, int32_t N, int32_t K,
                                    const float* W, int32_t Nf, int32_t trans_w,
                                    int64_t rows_per_batch, int64_t w_batch_stride, int32_t act,
                                    float* Y, float* AX, float* pre_act, int32_t grid_reserve,
                                    void* stream) {
  return tmgcn_spmm_gemm_f32_hint(rowptr, col, val, X, n_rows, N, K, W, Nf, trans_w, rows_per_batch, w_batch_stride,
                                  act, Y, AX, pre_act, grid_reserve, -1.f, stream);
}

#ifdef TMGCN_FUSED_TRACE
extern "C" int tmgcn_debug_fused_trace(unsigned long long* dst, long n_words, int clear) {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(tmgcn::fused_trace_words));
  if (e == hipSuccess && n_words > 0) e = hipMemcpy(dst, p, (size_t)n_words * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess && clear) e = hipMemset(p, 0, sizeof(unsigned long long) * 4096 * 16);
  return (int)e;
}
#endif

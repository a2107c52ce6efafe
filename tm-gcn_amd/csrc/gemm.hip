// P3 — feature·weight contraction and its two backward products   (gfx950 / CDNA4)
//
// Replaces  t.matmul(AtXt, Wt)  (embedding_help_functions.py:222, 330, 340, 344, 349, 415,
// 486-489) and autograd's  dA = dY·Wᵀ,  dW = Σ_r A[r]ᵀ dY[r].
//
//   gemm_bf16x3    Y[R][Nf] = act(A[R][K] · Wop) on the bf16 matrix cores after an exact 3-way split of
//                  the fp32 operands (fp32-accurate; the default for K a multiple of 4 in [16, 128]).
//   gemm_mfma      the same product as exact-f32 MFMA (v_mfma_f32_32x32x2_f32): every other shape, and on request.
//                  A 64-row tile is staged through LDS with full-line loads; each wave
//                  keeps its 32-column strip of W in registers (B fragments) for the whole
//                  persistent loop, so W is read once per block.  The k index inside a
//                  step is permuted (lane half h takes k = 8j+4h+s) so that one
//                  ds_read_b128 feeds four MFMAs.
//                  gemm_bf16x3<1> takes a weight STORED in bf16 as it is (three plane products).
//   gemm_small     thread-per-row FMA kernel for the reference's real sizes (2x6, 6x6, 12x2):
//                  MFMA tiles would be >90 % padding there; the op is a pure HBM stream
//                  (outputs leave through an LDS tile as one coalesced stream).
//   gemm_dw_narrow dW for even K, Nf <= 8: per-lane fp64 register sums, one block fold at the end;
//   gemm_dw_small  the general small-shape dW (LDS-staged rows).
//   gemm_dw_bf16x3 dW = AᵀdY on the bf16 matrix cores after an exact 3-way split of the fp32 operands
//                  (fp32-accurate, reproducible; the default for K, Nf >= 16) — see the kernel.
//   gemm_dw_lds    the exact-f32 MFMA form of the same product.  The reduction index is the row r, so the MFMA operands have the
//                  feature index on the lane; 32-row windows of A and dY are staged once per
//                  block through LDS (full-line loads) and read back with conflict-free
//                  ds_read_b32; row-chunk partial slabs are reduced in a fixed order by a
//                  second kernel (no float atomics: reproducible).
#include <type_traits>
#include "common.h"
#include "async_stage.h"

namespace tmgcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64;        // rows per tile
constexpr int KC = 128;       // k-chunk held in registers
constexpr int LDA = KC + 4;   // LDS row stride (floats): conflict-free ds_read_b128

struct GemmArgs {
  const float* A;
  const float* W;
  float* Y;
  float* pre;
  int64_t R;
  int32_t K, Nf;
  int32_t trans_w;
  int64_t rows_per_batch;  // 0 = shared W
  int64_t w_batch_stride;
  int32_t act;
  int64_t tiles_per_batch;
  int64_t n_tiles;
  unsigned int* tile_counter;  // dynamic tile scheduling (common.h); one counter per blockIdx.y strip
  int32_t w_bf16;              // W holds bf16 elements (tmgcn_gemm_bf16w_f32): W points at uint16_t
  int32_t stage_off;           // gemm_small: float offset of the output tile in dynamic LDS (0 = store directly)
  // gemm_bf16x3<WP, true> — one 128-wide k-chunk of a wider product (128 < K <= 512): K above is the
  // chunk's width, lda the full row length of A (and of a transposed W), k0 the chunk's first k;
  // accum: add to what Y already holds (every chunk but the first)
  int32_t lda, k0, accum;
};

// element (k, n) of the operator at element offset woff (the batch's weight) of W, fp32 or bf16 storage
__device__ __forceinline__ float wop(const GemmArgs& a, int64_t woff, int k, int n) {
  if (k >= a.K || n >= a.Nf) return 0.f;
  const int64_t i = woff + (a.trans_w ? (int64_t)n * a.K + k : (int64_t)k * a.Nf + n);
  if (a.w_bf16) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a.W)[i] << 16);
  return a.W[i];
}

__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmArgs a) {
  __shared__ float As[BM * LDA];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int n0 = blockIdx.y * 128 + wave * 32;  // this wave's column strip
  const int64_t batch_rows = a.rows_per_batch ? a.rows_per_batch : a.R;

  float wreg[KC / 8][4];
  int64_t cur_batch = -1;
  const int n_kchunks = (a.K + KC - 1) / KC;
  __shared__ unsigned int s_tile;

  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(a.tile_counter + blockIdx.y, 1u);
    __syncthreads();
    const int64_t tile = s_tile;
    if (tile >= a.n_tiles) break;
    const int64_t batch = tile / a.tiles_per_batch;
    const int64_t row0 = batch * batch_rows + (tile % a.tiles_per_batch) * BM;
    int64_t row_end = (batch + 1) * batch_rows;
    if (row_end > a.R) row_end = a.R;
    const int64_t Wb = a.rows_per_batch ? batch * a.w_batch_stride : 0;

    f32x16 acc[BM / 32];
#pragma unroll
    for (int mb = 0; mb < BM / 32; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;

    for (int kc = 0; kc < n_kchunks; ++kc) {
      const int k0 = kc * KC;
      int kw = a.K - k0;
      if (kw > KC) kw = KC;
      const int nj = (kw + 7) / 8;
      // B fragments: reload when the weight (batch) or the k-chunk changes
      if (batch != cur_batch || n_kchunks > 1) {
#pragma unroll
        for (int j = 0; j < KC / 8; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            wreg[j][s] = (j < nj) ? wop(a, Wb, k0 + 8 * j + 4 * lh + s, n0 + li) : 0.f;
      }
      __syncthreads();  // previous tile's fragment reads are done
      // stage A[row0 .. row0+BM) x [k0, k0+8*nj) into LDS, zero padded
      const int kw8 = nj * 8;
      if ((a.K % 4 == 0) && (reinterpret_cast<uintptr_t>(a.A) % 16 == 0)) {
        const int f4_per_row = kw8 / 4;
        for (int t = threadIdx.x; t < BM * f4_per_row; t += 256) {
          const int rr = t / f4_per_row, q = t % f4_per_row;
          const int64_t r = row0 + rr;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (r < row_end && k0 + 4 * q < a.K)
            v = *reinterpret_cast<const float4*>(a.A + r * a.K + k0 + 4 * q);
          *reinterpret_cast<float4*>(&As[rr * LDA + 4 * q]) = v;
        }
      } else {
        for (int t = threadIdx.x; t < BM * kw8; t += 256) {
          const int rr = t / kw8, q = t % kw8;
          const int64_t r = row0 + rr;
          As[rr * LDA + q] = (r < row_end && k0 + q < a.K) ? a.A[r * a.K + k0 + q] : 0.f;
        }
      }
      __syncthreads();
      if (n0 < a.Nf) {
#pragma unroll
        for (int mb = 0; mb < BM / 32; ++mb) {
#pragma unroll
          for (int j = 0; j < KC / 8; ++j) {
            if (j < nj) {
              const float4 av =
                  *reinterpret_cast<const float4*>(&As[(mb * 32 + li) * LDA + 8 * j + 4 * lh]);
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wreg[j][0], acc[mb], 0, 0, 0);
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wreg[j][1], acc[mb], 0, 0, 0);
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wreg[j][2], acc[mb], 0, 0, 0);
              acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wreg[j][3], acc[mb], 0, 0, 0);
            }
          }
        }
      }
    }
    cur_batch = batch;

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
    const int n = n0 + li;
    if (n < a.Nf) {
#pragma unroll
      for (int mb = 0; mb < BM / 32; ++mb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t r = row0 + mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
          if (r < row_end) {
            const float s = acc[mb][i];
            if (a.pre) a.pre[r * a.Nf + n] = s;
            a.Y[r * a.Nf + n] = act_apply(s, a.act);
          }
        }
      }
    }
    __syncthreads();  // everyone has read s_tile and the LDS tile before the next draw
  }
}

// ---- P3 on the bf16 matrix cores at fp32 accuracy ---------------------------------------------
// Y = act(A·Wop) with every fp32 operand split exactly into three bf16 planes (x = hi + mid + lo,
// split3 below) and the six plane products that matter per term, as in gemm_dw_bf16x3: 6 bf16 MFMAs
// (32 cycles each) do the work of 8 f32 MFMAs (64 cycles each), which moves the F = 128 GEMM from
// MFMA-bound (0.45 of the f32 peak in the kernel above) to its 2·R·128·4 B stream.
//   tile = 64 rows.  Thread (row group t>>5, quad t&31) loads rows (t>>5)+8i, i = 0..7, as coalesced
//   float4s (4 consecutive k), splits them and writes 8 bytes per plane into the LDS image
//   [plane][row: 272 B][k: 2 B]; the 16-byte row pad makes the ds_read_b128 A-fragment reads
//   (lane = row, 8 consecutive k) conflict-free.  The wave's 32-column strip of Wop lives in
//   registers as pre-split B fragments (8 k-steps x 3 planes x 4 VGPRs) for the whole persistent
//   loop.  The next tile's rows are requested before this tile's MFMAs (register staging), tiles
//   are drawn from the device counter one ahead.  The reduction is over K <= 128 only (48 MFMA
//   accumulations per output): the truncating bf16-MFMA accumulator (see X3_FLUSH) costs < 0.1 ulp here.
typedef __bf16 gx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gx_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned gx_pack(float a, float b) {  // bf16(a) | bf16(b) << 16, RNE
  gx_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gx_bf16x2));
}
__device__ __forceinline__ void gx_split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = gx_pack(a, b);
  a -= __uint_as_float(h << 16);
  b -= __uint_as_float(h & 0xffff0000u);
  m = gx_pack(a, b);
  a -= __uint_as_float(m << 16);
  b -= __uint_as_float(m & 0xffff0000u);
  l = gx_pack(a, b);
}

constexpr int GX_PITCH = 272;             // bytes per row of a plane image: 128 k x 2 B + 16 pad
constexpr int GX_PLANE = BM * GX_PITCH;   // 64 rows
typedef float gx_f32x4 __attribute__((ext_vector_type(4)));

struct GxTile {
  int64_t row0;
  int rows, batch;  // rows of the tile inside its batch (1..64)
};

// activation + stores of one 64-row tile.  C/D map of the 32x32 MFMA: col = lane&31,
// row = (i&3) + 8*(i>>2) + 4*(lane>>5).  One 64-bit base per lane and 32-bit row offsets; whole
// tiles take the branch-free path.
// ADD: the tile is one k-chunk's contribution — add what Y already holds before the activation
template <int ACT, bool ADD = false>
__device__ __forceinline__ void gx_store_rows(const f32x16 (&acc)[2], float* yb, float* pb, int Nf, int tile_rows,
                                              int rows_left) {
  if (tile_rows == BM) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int off = (mb * 32 + (i & 3) + 8 * (i >> 2)) * Nf;
        float sv = acc[mb][i];
        if (ADD) sv += yb[off];
        if (pb) pb[off] = sv;
        yb[off] = act_apply(sv, ACT);
      }
  } else {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = mb * 32 + (i & 3) + 8 * (i >> 2);
        if (rr < rows_left) {
          float sv = acc[mb][i];
          if (ADD) sv += yb[rr * Nf];
          if (pb) pb[rr * Nf] = sv;
          yb[rr * Nf] = act_apply(sv, ACT);
        }
      }
  }
}

// The same tile through 16-byte stores: each 4x4 block (4 rows in registers x 4 columns on a lane
// quad) is transposed in registers (quad_transpose4), after which lane j of a quad owns row j and
// columns 4q..4q+3.  y0 / p0 point at (tile row 0, this wave's column strip); needs Nf % 4 == 0 and
// 16-byte aligned outputs (checked by the caller).  Rows past the tile's end are predicated off.
template <int ACT, bool ADD = false>
__device__ __forceinline__ void gx_store_rows_v4(const f32x16 (&acc)[2], float* y0, float* p0, int Nf, int rows,
                                                 int li, int lh, bool cols_ok) {
  // opaque copies: otherwise the eight row offsets and row tests are hoisted out of the tile loop as
  // loop invariants and pinned in registers for the whole kernel, beside the 96-VGPR operator strip
  asm volatile("" : "+v"(li), "+v"(lh));
  const int j = li & 3, q = li >> 2;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4] = {acc[mb][4 * g], acc[mb][4 * g + 1], acc[mb][4 * g + 2], acc[mb][4 * g + 3]};
      quad_transpose4(v, j);
      const int rr = mb * 32 + 8 * g + 4 * lh + j;
      if (rr < rows && cols_ok) {
        const int off = rr * Nf + 4 * q;
        if (ADD) {
          const float4 old = *reinterpret_cast<const float4*>(y0 + off);
          v[0] += old.x;
          v[1] += old.y;
          v[2] += old.z;
          v[3] += old.w;
        }
        if (p0) *reinterpret_cast<float4*>(p0 + off) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(y0 + off) =
            make_float4(act_apply(v[0], ACT), act_apply(v[1], ACT), act_apply(v[2], ACT), act_apply(v[3], ACT));
      }
      __builtin_amdgcn_sched_barrier(0);  // one 4x4 block in flight: the B strip leaves ~60 VGPRs for everything else
    }
}

// WP = planes of the operator: 3 for an fp32 W (six plane products per term), 1 for a W stored in
// bf16 (tmgcn_gemm_bf16w_f32: the weight IS its high plane, three products per term, a third of the
// fragment registers) — the same sums in the same order, so bf16 W gives bit-identical results to
// its fp32-widened copy at half the matrix-core work.
// CH = the launch is one k-chunk of a product wider than 128 (see GemmArgs::lda): the non-chunked
// instantiations compile to exactly the code they had before the chunked form existed.
template <int WP, bool CH = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[3 * GX_PLANE];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int n0 = blockIdx.y * 128 + wave * 32;  // this wave's column strip
  const bool strip = n0 < a.Nf;
  const int64_t batch_rows = a.rows_per_batch ? a.rows_per_batch : a.R;
  const int nks = (a.K + 15) / 16;  // k-steps of 16

  // staging role: quad q (k = 4q .. 4q+3) of rows rg + 8 i
  const int q = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const bool okq = 4 * q < a.K;             // K % 4 == 0: a quad is inside or outside as a whole
  const float zq = okq ? 1.f : 0.f;
  const int qcol = okq ? 4 * q : 0;         // outside quads read column 0 (a valid address) and are zeroed
  unsigned char* wr = sm + rg * GX_PITCH + q * 8;
  const unsigned char* rd = sm + li * GX_PITCH + lh * 16;

  unsigned bw[8][WP][4];  // B fragments of Wop: [k-step][plane][8 bf16]
  // Wop[k][n] = W[k*sk + n*sn]; this lane's column n0+li, clamped (columns / rows outside are zeroed by
  // the mask, so the 64 loads are unconditional: no exec-mask branches)
  const int ncol = n0 + li < a.Nf ? n0 + li : a.Nf - 1;
  const int LD = CH ? a.lda : a.K;  // row length of A / of a transposed W in memory
  const int sk = a.trans_w ? 1 : a.Nf, sn = a.trans_w ? LD : 1;
  const float zn = n0 + li < a.Nf ? 1.f : 0.f;
  auto load_w = [&](int64_t woff) __attribute__((always_inline)) {
    int koff = 8 * lh;
    // opaque to the optimiser: otherwise the 64 loop-invariant element offsets are hoisted out of
    // the tile loop into 64 VGPRs (and spilled) for a routine that runs once per weight
    asm volatile("" : "+v"(koff));
    const int64_t wl = woff + (int64_t)ncol * sn + (CH ? (int64_t)a.k0 * sk : 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = koff + 16 * ks + 2 * j;
        const int k0c = k < a.K ? k : a.K - 1, k1c = k + 1 < a.K ? k + 1 : a.K - 1;
        if constexpr (WP == 3) {
          const float w0 = a.W[wl + (int64_t)k0c * sk] * (k < a.K ? zn : 0.f);
          const float w1 = a.W[wl + (int64_t)k1c * sk] * (k + 1 < a.K ? zn : 0.f);
          gx_split3(w0, w1, bw[ks][0][j], bw[ks][1][j], bw[ks][2][j]);
        } else {
          const unsigned short* wh = reinterpret_cast<const unsigned short*>(a.W);
          const unsigned w0 = (k < a.K && zn != 0.f) ? wh[wl + (int64_t)k0c * sk] : 0u;
          const unsigned w1 = (k + 1 < a.K && zn != 0.f) ? wh[wl + (int64_t)k1c * sk] : 0u;
          bw[ks][0][j] = w0 | (w1 << 16);
        }
      }
  };
  // the plane products of one (k-step, row block): small terms first
  auto products = [&](f32x16& c, const gx_bf16x8 ah, const gx_bf16x8 am, const gx_bf16x8 al, int ks) __attribute__((always_inline)) {
    const gx_bf16x8 bh = __builtin_bit_cast(gx_bf16x8, make_uint4(bw[ks][0][0], bw[ks][0][1], bw[ks][0][2], bw[ks][0][3]));
    if constexpr (WP == 3) {
      const gx_bf16x8 bm = __builtin_bit_cast(gx_bf16x8, make_uint4(bw[ks][1][0], bw[ks][1][1], bw[ks][1][2], bw[ks][1][3]));
      const gx_bf16x8 bl = __builtin_bit_cast(gx_bf16x8, make_uint4(bw[ks][2][0], bw[ks][2][1], bw[ks][2][2], bw[ks][2][3]));
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    }
  };

  // Two staging sets of 8 float4s in RESERVED registers (v192..v255, async_stage.h), filled by
  // inline-asm loads that the compiler's wait-count pass does not see: the rows of tile i+2 are
  // requested as soon as tile i has been split and are waited for with a COUNTED s_waitcnt one tile
  // later (vmcnt(8): everything but the 8 loads requested since), so ~2 x 32 KB per block stay in
  // flight across the barriers.  With compiler-visible loads hipcc drains the queue (vmcnt(0)) at
  // the first use after the branches of the epilogue, which degrades a two-deep ring to a one-deep
  // one (measured in round 1 on the dW kernel).
  auto tile_rows = [&](unsigned tile) __attribute__((always_inline)) {  // tile ids fit 31 bits (checked by the launcher): 32-bit scalar arithmetic
    const unsigned tpb = (unsigned)a.tiles_per_batch;
    const unsigned b = tile / tpb;
    GxTile t;
    t.batch = (int)b;
    t.row0 = (int64_t)b * batch_rows + (int64_t)(tile - b * tpb) * BM;
    int64_t row_end = (int64_t)(b + 1) * batch_rows;
    if (row_end > a.R) row_end = a.R;
    const int64_t left = row_end - t.row0;
    t.rows = left < BM ? (int)left : BM;
    return t;
  };
  // scalar base of the tile + a 32-bit per-lane offset: no 64-bit address pairs held in VGPRs.
  // No conditional load: rows past the end re-read the tile's last row and are zeroed at the split.
  auto fetch = [&](auto set, const GxTile& t) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const float* tbase = a.A + t.row0 * LD + (CH ? a.k0 : 0);
    auto voff = [&](int i) __attribute__((always_inline)) {
      int rr = rg + 8 * i;
      if (rr >= t.rows) rr = t.rows - 1;
      return (unsigned)(rr * LD + qcol) * 4u;
    };
    stage8_load_s<SET, 0>(voff(0), tbase);
    stage8_load_s<SET, 1>(voff(1), tbase);
    stage8_load_s<SET, 2>(voff(2), tbase);
    stage8_load_s<SET, 3>(voff(3), tbase);
    stage8_load_s<SET, 4>(voff(4), tbase);
    stage8_load_s<SET, 5>(voff(5), tbase);
    stage8_load_s<SET, 6>(voff(6), tbase);
    stage8_load_s<SET, 7>(voff(7), tbase);
  };
  auto landed = [&](bool newer_in_flight) __attribute__((always_inline)) {  // the OLDER set's 8 loads are complete
    if (newer_in_flight)
      TMGCN_WAIT_VM(8);
    else
      TMGCN_WAIT_VM(0);
  };
  auto zrow = [&](int i, const GxTile& t) __attribute__((always_inline)) { return (rg + 8 * i < t.rows) ? zq : 0.f; };  // 0 for rows / k-quads outside
  auto split_one = [&](const stage_f32x4& v, int i) __attribute__((always_inline)) {  // v arrives masked (stage8_read_mul)
    unsigned h0, m0, l0, h1, m1, l1;
    gx_split3(v[0], v[1], h0, m0, l0);
    gx_split3(v[2], v[3], h1, m1, l1);
    unsigned char* w = wr + i * 8 * GX_PITCH;
    *reinterpret_cast<uint2*>(w) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(w + GX_PLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(w + 2 * GX_PLANE) = make_uint2(l0, l1);
  };
  auto split_store = [&](auto set, const GxTile& t) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    split_one(stage8_read_mul<SET, 0>(zrow(0, t)), 0);
    split_one(stage8_read_mul<SET, 1>(zrow(1, t)), 1);
    split_one(stage8_read_mul<SET, 2>(zrow(2, t)), 2);
    split_one(stage8_read_mul<SET, 3>(zrow(3, t)), 3);
    split_one(stage8_read_mul<SET, 4>(zrow(4, t)), 4);
    split_one(stage8_read_mul<SET, 5>(zrow(5, t)), 5);
    split_one(stage8_read_mul<SET, 6>(zrow(6, t)), 6);
    split_one(stage8_read_mul<SET, 7>(zrow(7, t)), 7);
  };
  const std::integral_constant<int, 0> SA;
  const std::integral_constant<int, 1> SB;

  // the tile whose planes are in LDS: multiply() runs the MFMAs, store_tile() applies the activation
  // and stores; the counted wait for the other staging set sits between the two
  int cur_batch = -1;
  f32x16 acc[2];
  auto k_step = [&](int ks) __attribute__((always_inline)) {  // 2 row blocks x 6 plane products of one 16-deep k-step
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const unsigned char* p = rd + mb * 32 * GX_PITCH + ks * 32;
      const gx_bf16x8 ah = __builtin_bit_cast(gx_bf16x8, *reinterpret_cast<const uint4*>(p));
      const gx_bf16x8 am = __builtin_bit_cast(gx_bf16x8, *reinterpret_cast<const uint4*>(p + GX_PLANE));
      const gx_bf16x8 al = __builtin_bit_cast(gx_bf16x8, *reinterpret_cast<const uint4*>(p + 2 * GX_PLANE));
      products(acc[mb], ah, am, al, ks);
    }
  };
  auto multiply = [&](const GxTile& tc) __attribute__((always_inline)) {
    if (tc.batch != cur_batch) {
      load_w(a.rows_per_batch ? (int64_t)tc.batch * a.w_batch_stride : 0);
      cur_batch = tc.batch;
    }
    if (!strip) return;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
    if (nks == 8) {
      // K = 128: 16 groups (k-step, row block) of 3 fragment reads + 6 MFMAs, software-pipelined by
      // hand: the reads of group g+1 are issued in front of the MFMAs of group g (sched_barrier pins
      // the order), so a read has 192 cycles of matrix-pipe time to land.  Left alone, hipcc issues
      // each read right in front of its MFMA and waits out the LDS latency ~50 times per tile.
      struct Frag3 {
        uint4 h, m, l;
      };
      auto read_group = [&](int g) __attribute__((always_inline)) {
        const unsigned char* p = rd + (g & 1) * 32 * GX_PITCH + (g >> 1) * 32;
        Frag3 f;
        f.h = *reinterpret_cast<const uint4*>(p);
        f.m = *reinterpret_cast<const uint4*>(p + GX_PLANE);
        f.l = *reinterpret_cast<const uint4*>(p + 2 * GX_PLANE);
        return f;
      };
      Frag3 cur = read_group(0);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int ks = g >> 1, mb = g & 1;
        Frag3 nxt = cur;
        if (g + 1 < 16) nxt = read_group(g + 1);
        __builtin_amdgcn_sched_barrier(0);
        const gx_bf16x8 ah = __builtin_bit_cast(gx_bf16x8, cur.h), am = __builtin_bit_cast(gx_bf16x8, cur.m),
                        al = __builtin_bit_cast(gx_bf16x8, cur.l);
        products(acc[mb], ah, am, al, ks);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        if (ks < nks) k_step(ks);
    }
  };
  // 16-byte stores need whole column quads inside Nf and aligned rows
  const bool v4 = (a.Nf % 4 == 0) && (reinterpret_cast<uintptr_t>(a.Y) % 16 == 0) &&
                  (!a.pre || reinterpret_cast<uintptr_t>(a.pre) % 16 == 0);
  auto store_tile = [&](const GxTile& tc) __attribute__((always_inline)) {
    if (!strip) return;
    if (v4) {  // every lane takes part in the quad transposes; columns past Nf are predicated off
      const int64_t base = tc.row0 * a.Nf + n0;
      float* y0 = a.Y + base;
      float* p0 = a.pre ? a.pre + base : nullptr;
      const bool cols_ok = n0 + 4 * (li >> 2) < a.Nf;
      if constexpr (CH) {
        if (a.accum) {
          switch (a.act) {
            case TMGCN_ACT_RELU: gx_store_rows_v4<TMGCN_ACT_RELU, true>(acc, y0, p0, a.Nf, tc.rows, li, lh, cols_ok); break;
            case TMGCN_ACT_LEAKY: gx_store_rows_v4<TMGCN_ACT_LEAKY, true>(acc, y0, p0, a.Nf, tc.rows, li, lh, cols_ok); break;
            case TMGCN_ACT_SELU: gx_store_rows_v4<TMGCN_ACT_SELU, true>(acc, y0, p0, a.Nf, tc.rows, li, lh, cols_ok); break;
            default: gx_store_rows_v4<TMGCN_ACT_NONE, true>(acc, y0, p0, a.Nf, tc.rows, li, lh, cols_ok);
          }
          return;
        }
      }
      switch (a.act) {
        case TMGCN_ACT_RELU: gx_store_rows_v4<TMGCN_ACT_RELU>(acc, y0, p0, a.Nf, tc.rows, li, lh, cols_ok); break;
        case TMGCN_ACT_LEAKY: gx_store_rows_v4<TMGCN_ACT_LEAKY>(acc, y0, p0, a.Nf, tc.rows, li, lh, cols_ok); break;
        case TMGCN_ACT_SELU: gx_store_rows_v4<TMGCN_ACT_SELU>(acc, y0, p0, a.Nf, tc.rows, li, lh, cols_ok); break;
        default: gx_store_rows_v4<TMGCN_ACT_NONE>(acc, y0, p0, a.Nf, tc.rows, li, lh, cols_ok);
      }
      return;
    }
    const int n = n0 + li;
    if (n >= a.Nf) return;
    const int64_t base = (tc.row0 + 4 * lh) * a.Nf + n;
    float* yb = a.Y + base;
    float* pb = a.pre ? a.pre + base : nullptr;
    const int rows_left = tc.rows - 4 * lh;
    if constexpr (CH) {
      if (a.accum) {
        switch (a.act) {
          case TMGCN_ACT_RELU: gx_store_rows<TMGCN_ACT_RELU, true>(acc, yb, pb, a.Nf, tc.rows, rows_left); break;
          case TMGCN_ACT_LEAKY: gx_store_rows<TMGCN_ACT_LEAKY, true>(acc, yb, pb, a.Nf, tc.rows, rows_left); break;
          case TMGCN_ACT_SELU: gx_store_rows<TMGCN_ACT_SELU, true>(acc, yb, pb, a.Nf, tc.rows, rows_left); break;
          default: gx_store_rows<TMGCN_ACT_NONE, true>(acc, yb, pb, a.Nf, tc.rows, rows_left);
        }
        return;
      }
    }
    switch (a.act) {  // chosen once per tile, not once per element
      case TMGCN_ACT_RELU: gx_store_rows<TMGCN_ACT_RELU>(acc, yb, pb, a.Nf, tc.rows, rows_left); break;
      case TMGCN_ACT_LEAKY: gx_store_rows<TMGCN_ACT_LEAKY>(acc, yb, pb, a.Nf, tc.rows, rows_left); break;
      case TMGCN_ACT_SELU: gx_store_rows<TMGCN_ACT_SELU>(acc, yb, pb, a.Nf, tc.rows, rows_left); break;
      default: gx_store_rows<TMGCN_ACT_NONE>(acc, yb, pb, a.Nf, tc.rows, rows_left);
    }
  };

  // Static persistent schedule: block b takes tiles b, b + G, b + 2G, ... (uniform tiles: nothing to
  // balance, and a returning atomic would be one more asynchronous result to park in a reserved
  // register).  Tiles alternate between the two staging sets, so the loop body is written for two.
  const unsigned n_tiles = (unsigned)a.n_tiles, stride = gridDim.x;
  unsigned t0 = blockIdx.x, t1 = blockIdx.x + stride;
  GxTile ta = tile_rows(t0 < n_tiles ? t0 : 0), tb = tile_rows(t1 < n_tiles ? t1 : 0);
  if (t0 < n_tiles) fetch(SA, ta);
  if (t1 < n_tiles) fetch(SB, tb);
  landed(t1 < n_tiles);  // set A
  while (t0 < n_tiles) {
    // ---- tile t0 (set A): split, request t0 + 2G into set A, multiply, wait for set B, store
    __syncthreads();  // the previous tile's fragment reads are done
    split_store(SA, ta);
    const unsigned t2 = t0 + 2 * stride;
    const bool f2 = t2 < n_tiles;
    const GxTile tc2 = tile_rows(f2 ? t2 : 0);
    if (f2) fetch(SA, tc2);
    __syncthreads();
    multiply(ta);
    landed(f2);  // set B
    store_tile(ta);
    if (t1 >= n_tiles) break;
    // ---- tile t1 (set B)
    __syncthreads();
    split_store(SB, tb);
    const unsigned t3 = t1 + 2 * stride;
    const bool f3 = t3 < n_tiles;
    const GxTile tc3 = tile_rows(f3 ? t3 : 0);
    if (f3) fetch(SB, tc3);
    __syncthreads();
    multiply(tb);
    landed(f3);  // set A
    store_tile(tb);
    t0 = t2;
    ta = tc2;
    t1 = t3;
    tb = tc3;
  }
}

// thread-per-row kernel for small K / Nf; W (zero padded to a multiple of 8 columns) in LDS.
// With stage_off the block's 256 x Nf outputs go through an LDS tile and leave as one contiguous,
// fully coalesced stream (a row is Nf*4 bytes: stored lane by lane it is Nf dword stores at that
// stride per wave — 21 us for 570 k rows of 2 -> 6 with the pre-activation, against 8 us of bytes).
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs a) {
  extern __shared__ float Ws[];  // [n_w][K][Nfp] (+ [256][Nfp+1] output tile at stage_off)
  const ActApply act(a.act);    // decoded once: no switch per element (common.h)
  const int Nfp = (a.Nf + 7) & ~7;
  const int64_t batch_rows = a.rows_per_batch ? a.rows_per_batch : a.R;
  const int64_t r_first = (int64_t)blockIdx.x * blockDim.x;
  int64_t r_last = r_first + blockDim.x - 1;
  if (r_last >= a.R) r_last = a.R - 1;
  const int64_t b_first = r_first / batch_rows;
  const int n_w = (int)(r_last / batch_rows - b_first) + 1;  // weights this block touches
  for (int t = threadIdx.x; t < n_w * a.K * Nfp; t += blockDim.x) {
    const int wb = t / (a.K * Nfp), rem = t % (a.K * Nfp);
    const int k = rem / Nfp, n = rem % Nfp;
    Ws[t] = wop(a, a.rows_per_batch ? (b_first + wb) * a.w_batch_stride : 0, k, n);
  }
  __syncthreads();
  const int64_t r = r_first + threadIdx.x;
  const bool live = r < a.R;
  float* tile = a.stage_off ? Ws + a.stage_off + threadIdx.x * (Nfp + 1) : nullptr;
  if (live) {
    const float* Wl = Ws + (int64_t)(r / batch_rows - b_first) * a.K * Nfp;
    const float* Ar = a.A + r * a.K;
    for (int nb = 0; nb < Nfp; nb += 8) {
      float acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      for (int k = 0; k < a.K; ++k) {
        const float x = Ar[k];
        const float* w = Wl + k * Nfp + nb;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fmaf(x, w[i], acc[i]);
      }
      if (tile) {
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[nb + i] = acc[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (nb + i < a.Nf) {
            if (a.pre) a.pre[r * a.Nf + nb + i] = acc[i];
            a.Y[r * a.Nf + nb + i] = act(acc[i]);
          }
        }
      }
    }
  }
  if (!a.stage_off) return;
  __syncthreads();
  const float* t0 = Ws + a.stage_off;
  const int total = (int)(r_last - r_first + 1) * a.Nf;
  float* yb = a.Y + r_first * a.Nf;
  float* pb = a.pre ? a.pre + r_first * a.Nf : nullptr;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int row = idx / a.Nf, n = idx - row * a.Nf;
    const float v = t0[row * (Nfp + 1) + n];
    if (pb) pb[idx] = v;
    yb[idx] = act(v);
  }
}

// --------------------------------------------------------------------------------------
// dW
// --------------------------------------------------------------------------------------
struct DwArgs {
  const float* A;
  const float* dY;
  float* part;  // [n_batch][chunks][K][Nf]
  int64_t R;
  int32_t K, Nf;
  int64_t batch_rows;
  int32_t chunks;  // row chunks per batch
  int64_t rows_per_chunk;
  float* dW = nullptr;       // narrow kernel: the last block to finish reduces the slabs into dW itself
  int32_t* sync = nullptr;   //   (hand-off word: zero on entry, left zero)
  int32_t n_batch = 1;
  const float* pre = nullptr;  // narrow kernel, optional: dY is multiplied by act'(pre) on the fly
  int32_t act = 0;
};

// grid: x = batch*chunks + chunk, y = 128x128 output tile (ky * n_tiles_n + ny)
// LDS-staged dW: the block stages DWT rows of A and dY (the 128-column windows of its output
// tile) once with full-line loads; all four waves then read their MFMA operands from LDS with
// conflict-free ds_read_b32 (lane = feature).  Without the staging every wave re-reads the same
// dY rows through L1 (4x redundancy) and the kernel sits at 0.55 of the f32 MFMA rate, sensitive
// to how many loads are in flight (deeper unrolling made it slower: L1 thrash).
#ifndef TMGCN_DWT
#define TMGCN_DWT 32
#endif
constexpr int DWT = TMGCN_DWT;  // rows per LDS tile: 32 x (128 + 128) x 4 B = 32 KB
__global__ __launch_bounds__(256) void gemm_dw_lds_kernel(DwArgs a) {
  __shared__ float sA[DWT * 128];
  __shared__ float sB[DWT * 128];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int tiles_n = (a.Nf + 127) / 128;
  const int kt = blockIdx.y / tiles_n, nt = blockIdx.y % tiles_n;
  const int64_t batch = blockIdx.x / a.chunks;
  const int chunk = blockIdx.x % a.chunks;
  const int64_t b0 = batch * a.batch_rows;
  int64_t b1 = b0 + a.batch_rows;
  if (b1 > a.R) b1 = a.R;
  const int64_t r0 = b0 + (int64_t)chunk * a.rows_per_chunk;
  int64_t r1 = r0 + a.rows_per_chunk;
  if (r1 > b1) r1 = b1;
  const int kbase = kt * 128, nbase = nt * 128;
  const bool vecA = a.K % 4 == 0 && (reinterpret_cast<uintptr_t>(a.A) % 16 == 0);
  const bool vecB = a.Nf % 4 == 0 && (reinterpret_cast<uintptr_t>(a.dY) % 16 == 0);

  f32x16 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

  // (a register-prefetched variant of this loop measured the same 92.5 TF at 3 instead of 4 blocks
  // per CU: with four blocks resident the staging of one overlaps the MFMAs of the others)
  for (int64_t r = r0; r < r1; r += DWT) {
    const int nr = (int)((r1 - r) < DWT ? (r1 - r) : DWT);
    __syncthreads();  // previous tile consumed
    // stage [nr][128] windows of A and dY (zero padded in rows and columns)
    for (int t = threadIdx.x; t < DWT * 32; t += 256) {
      const int i = t >> 5, q = (t & 31) * 4;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (i < nr) {
        const float* ga = a.A + (r + i) * a.K + kbase + q;
        const float* gb = a.dY + (r + i) * a.Nf + nbase + q;
        if (vecA && kbase + q + 3 < a.K) va = *reinterpret_cast<const float4*>(ga);
        else {
          if (kbase + q < a.K) va.x = ga[0];
          if (kbase + q + 1 < a.K) va.y = ga[1];
          if (kbase + q + 2 < a.K) va.z = ga[2];
          if (kbase + q + 3 < a.K) va.w = ga[3];
        }
        if (vecB && nbase + q + 3 < a.Nf) vb = *reinterpret_cast<const float4*>(gb);
        else {
          if (nbase + q < a.Nf) vb.x = gb[0];
          if (nbase + q + 1 < a.Nf) vb.y = gb[1];
          if (nbase + q + 2 < a.Nf) vb.z = gb[2];
          if (nbase + q + 3 < a.Nf) vb.w = gb[3];
        }
      }
      *reinterpret_cast<float4*>(&sA[i * 128 + q]) = va;
      *reinterpret_cast<float4*>(&sB[i * 128 + q]) = vb;
    }
    __syncthreads();
    // operands of 4 row pairs are read from LDS one group ahead of the 16 MFMAs that use them
    // (left to itself hipcc emits ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs with one register
    // pair, exposing the LDS latency 32 times per tile)
    constexpr int GS = 2;                 // row pairs per group (4 costs a wave of occupancy: 136 VGPRs)
    constexpr int NG = DWT / 2 / GS;      // groups per tile
    float av[2][GS], bv[2][GS][4];
    const float* pA = sA + lh * 128 + wave * 32 + li;
    const float* pB = sB + lh * 128 + li;
#pragma unroll
    for (int u = 0; u < GS; ++u) {
      av[0][u] = pA[(2 * u) * 128];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) bv[0][u][nb] = pB[(2 * u) * 128 + nb * 32];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g + 1 < NG) {
#pragma unroll
        for (int u = 0; u < GS; ++u) {
          av[nxt][u] = pA[(2 * ((g + 1) * GS + u)) * 128];
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) bv[nxt][u][nb] = pB[(2 * ((g + 1) * GS + u)) * 128 + nb * 32];
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the next group's reads ahead of this group's MFMAs
#pragma unroll
      for (int u = 0; u < GS; ++u)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][u], bv[cur][u][nb], acc[nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* P = a.part + ((int64_t)blockIdx.x) * a.K * a.Nf;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = nbase + nb * 32 + li;
    if (n < a.Nf) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = kbase + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        if (kk < a.K) P[(int64_t)kk * a.Nf + n] = acc[nb][i];
      }
    }
  }
}

// dW on the bf16 matrix cores at fp32 accuracy: every fp32 operand is split exactly into three
// bf16 planes  x = hi + mid + lo  (8 + 8 + 8 mantissa bits; each residual is exact in fp32), and
// a product is the six plane products that matter,
//     a·b ≈ hi·hi + (hi·mid + mid·hi) + (hi·lo + lo·hi + mid·mid)          (dropped: ≤ 2^-24 |a·b|)
// each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs do the work of
// eight f32 ones at 16x the rate, which turns dW from MFMA-bound (0.62 of the f32 peak) into a
// stream over its two operands.
//   step = 32 rows.  Thread (kg, fq) loads rows 4kg..4kg+3 x features 4fq..4fq+3 of both operand
//   windows (coalesced float4s), splits them, and writes for each feature the four k-slots it owns
//   as one 8-byte store into the LDS image  [operand][plane][feature quad: 272 B][feature: 64 B][slot: 2 B];
//   the 16-byte pad per quad makes the ds_read_b128 fragment reads conflict-free (2-way on the
//   stores, which their issue cost hides).  k-slot = row inside the step for BOTH operands, so the
//   MFMA's k order is consistent.  Staging registers are refilled for the next step before the MFMA
//   phase; two blocks per CU (the RNE flush set below costs the third) overlap one block's split/store
//   phase with the other's MFMAs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // bf16(a) | bf16(b) << 16, RNE
  f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// (a, b) -> the three packed planes
__device__ __forceinline__ void split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pack_bf16(a, b);
  a -= bf16_lo(h);
  b -= bf16_hi(h);
  m = pack_bf16(a, b);
  a -= bf16_lo(m);
  b -= bf16_hi(m);
  l = pack_bf16(a, b);
}

// Measured in round 1, before the flush set (33.5 M rows, 128x128): depth 1 at 3 blocks/CU 7.57 ms, depth 2 / 3 / 4 at 2 blocks/CU 7.68 /
// 7.69 / 7.73 ms — hipcc drains every outstanding load at the ring loop's header (s_waitcnt vmcnt(0)),
// so a deeper ring buys nothing today; the f32-MFMA kernel it replaces takes 10.9 ms.
#ifndef X3_DEPTH
#define X3_DEPTH 1
#endif
#ifndef X3_OCC
#define X3_OCC 2
#endif
// v_mfma_f32_32x32x16_bf16 does not round its accumulation to nearest: the sum of the 16 products and
// the C input is TRUNCATED toward -infinity a few (~9-10) bits below the ulp of the largest addend
// (measured, tools/dw_bias.py: mean signed error of all 128x128 outputs < 0 for every operand
// distribution, growing like R·ulp(|acc|), while the f32 MFMA — an RNE fmaf chain — shows none).
// At the bench size (R = 33.5 M rows, 21.8 k rows per block) that bias reached 1.2e-5 of max|dW|,
// above the stated 1e-5 tolerance.  So the MFMA accumulator is kept SMALL: every X3_FLUSH steps
// (X3_FLUSH*32 rows) it is added into a second register set with v_add_f32 (round to nearest even)
// and cleared; the truncation then happens at ulp(|acc| <~ 8) instead of ulp(~100).
#ifndef X3_FLUSH
#define X3_FLUSH 4
#endif
#ifndef X3_SCHED
#define X3_SCHED 1
#endif
// (The same products on v_mfma_f32_16x16x32_bf16 — 16 tiles of 16x16 per wave and step — were measured in round 3: the same
// time within 2 %, a higher clock at twice the MFMA count; profiles/archive/r3c_ab_dw_m16.txt.  Not kept.)
constexpr int X3_ROWS = 32;
constexpr int X3_PITCH = 272;             // bytes per feature quad (4 x 64 + 16)
constexpr int X3_PLANE = 32 * X3_PITCH;   // 128 features
constexpr int X3_OPERAND = 3 * X3_PLANE;

__device__ __forceinline__ float comp(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

__global__ __launch_bounds__(256, X3_OCC) void gemm_dw_bf16x3_kernel(DwArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[2 * X3_OPERAND];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int tiles_n = (a.Nf + 127) / 128;
  const int kt = blockIdx.y / tiles_n, nt = blockIdx.y % tiles_n;
  const int64_t batch = blockIdx.x / a.chunks;
  const int chunk = blockIdx.x % a.chunks;
  const int64_t b0 = batch * a.batch_rows;
  int64_t b1 = b0 + a.batch_rows;
  if (b1 > a.R) b1 = a.R;
  const int64_t r0 = b0 + (int64_t)chunk * a.rows_per_chunk;
  int64_t r1 = r0 + a.rows_per_chunk;
  if (r1 > b1) r1 = b1;
  const int kbase = kt * 128, nbase = nt * 128;

  // staging role of this thread
  const int kg = 2 * wave + lh;  // rows 4kg .. 4kg+3 of the step
  const int fq = li;             // features 4fq .. 4fq+3 of the 128-wide windows
  // K and Nf are multiples of 4, so a feature quad is inside or outside as a whole; a thread whose
  // quad is outside reads column 0 instead (a valid address) and zeroes the values: the steady state
  // has no conditional load (those force s_waitcnt vmcnt(0) and exec-mask branches)
  const bool okA = kbase + 4 * fq < a.K, okB = nbase + 4 * fq < a.Nf;
  const float* pa = a.A + (r0 + 4 * kg) * a.K + (okA ? kbase + 4 * fq : 0);
  const float* pb = a.dY + (r0 + 4 * kg) * a.Nf + (okB ? nbase + 4 * fq : 0);
  const float za = okA ? 1.f : 0.f, zb = okB ? 1.f : 0.f;
  // X3_DEPTH staging sets form a ring: the rows of step n+DEPTH are requested as soon as step n has
  // been split, so a request has DEPTH whole steps to arrive.  The ring loop contains no conditional
  // load (the compiler then counts outstanding loads exactly: s_waitcnt vmcnt(8*(DEPTH-1)) instead of
  // draining every request before each split); chunk heads/tails go through the plain loop below.
  float4 sa[X3_DEPTH][4], sb[X3_DEPTH][4];
  auto fetch_full = [&](float4 (&xa)[4], float4 (&xb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xa[i] = *reinterpret_cast<const float4*>(pa + (int64_t)i * a.K);
      xb[i] = *reinterpret_cast<const float4*>(pb + (int64_t)i * a.Nf);
    }
    pa += (int64_t)X3_ROWS * a.K;
    pb += (int64_t)X3_ROWS * a.Nf;
  };
  auto fetch_any = [&](float4 (&xa)[4], float4 (&xb)[4], int64_t r) {  // r < r1; rows past r1 read row r1-1 and are zeroed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = r + 4 * kg + i;
      const int64_t back = row < r1 ? 0 : row - (r1 - 1);
      const float z = row < r1 ? 1.f : 0.f;
      const float4 va = *reinterpret_cast<const float4*>(pa + ((int64_t)i - back) * a.K);
      const float4 vb = *reinterpret_cast<const float4*>(pb + ((int64_t)i - back) * a.Nf);
      xa[i] = make_float4(va.x * z, va.y * z, va.z * z, va.w * z);
      xb[i] = make_float4(vb.x * z, vb.y * z, vb.z * z, vb.w * z);
    }
    pa += (int64_t)X3_ROWS * a.K;
    pb += (int64_t)X3_ROWS * a.Nf;
  };

  f32x16 acc[4], sum[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = sum[t][i] = 0.f;
  int since_flush = 0;
  auto flush = [&]() {  // RNE add of the short MFMA chain into the running sum (see X3_FLUSH)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sum[t][i] += acc[t][i];
        acc[t][i] = 0.f;
      }
    since_flush = 0;
  };

  unsigned char* wrA = sm + fq * X3_PITCH + kg * 8;
  unsigned char* wrB = wrA + X3_OPERAND;
  // fragment addresses: feature 32*tile + li -> quad 8*tile + li/4, feature-in-quad li%4; k half lh
  const unsigned char* rdA = sm + (8 * wave + (li >> 2)) * X3_PITCH + (li & 3) * 64 + lh * 16;
  const unsigned char* rdB = sm + X3_OPERAND + (li >> 2) * X3_PITCH + (li & 3) * 64 + lh * 16;

  auto split_store = [&](const float4 (&xa)[4], const float4 (&xb)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned h0, m0, l0, h1, m1, l1;
      split3(comp(xa[0], c) * za, comp(xa[1], c) * za, h0, m0, l0);
      split3(comp(xa[2], c) * za, comp(xa[3], c) * za, h1, m1, l1);
      *reinterpret_cast<uint2*>(wrA + c * 64) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(wrA + X3_PLANE + c * 64) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(wrA + 2 * X3_PLANE + c * 64) = make_uint2(l0, l1);
      split3(comp(xb[0], c) * zb, comp(xb[1], c) * zb, h0, m0, l0);
      split3(comp(xb[2], c) * zb, comp(xb[3], c) * zb, h1, m1, l1);
      *reinterpret_cast<uint2*>(wrB + c * 64) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(wrB + X3_PLANE + c * 64) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(wrB + 2 * X3_PLANE + c * 64) = make_uint2(l0, l1);
    }
  };
  auto multiply = [&]() {
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(rdA + sh * 32));
      const bf16x8 am = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(rdA + X3_PLANE + sh * 32));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(rdA + 2 * X3_PLANE + sh * 32));
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const unsigned char* q = rdB + t * 8 * X3_PITCH + sh * 32;
        const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q));
        const bf16x8 bm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + X3_PLANE));
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + 2 * X3_PLANE));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);  // small terms first
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
      }
    }
#if X3_SCHED
    // Software-pipeline the LDS fragment reads against the MFMAs (left to itself hipcc issues each
    // group's ds_read_b128s right in front of the MFMA that needs them and waits out their latency,
    // 8-10 times per step): A fragments of half 0 + B fragments of group 0 first, then every group of
    // 6 MFMAs is preceded by the 3 reads of the NEXT group (the A fragments of half 1 ride along with
    // group 2), so a read has 6 MFMAs = 192 cycles of matrix-pipe time to land.
#define X3_SGB_READS(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define X3_SGB_MFMAS(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
    X3_SGB_READS(6);                   // A(half 0), B(group 0)
    X3_SGB_READS(3); X3_SGB_MFMAS(6);  // B(1) | group 0
    X3_SGB_READS(3); X3_SGB_MFMAS(6);  // B(2) | group 1
    X3_SGB_READS(6); X3_SGB_MFMAS(6);  // B(3) + A(half 1) | group 2
    X3_SGB_READS(3); X3_SGB_MFMAS(6);  // B(4) | group 3
    X3_SGB_READS(3); X3_SGB_MFMAS(6);  // B(5) | group 4
    X3_SGB_READS(3); X3_SGB_MFMAS(6);  // B(6) | group 5
    X3_SGB_READS(3); X3_SGB_MFMAS(6);  // B(7) | group 6
    X3_SGB_MFMAS(6);                   //        group 7
#undef X3_SGB_READS
#undef X3_SGB_MFMAS
#endif
    if (++since_flush == X3_FLUSH) flush();
  };

  int64_t r = r0;
  const int64_t full_end = r0 + ((r1 - r0) / X3_ROWS) * X3_ROWS;  // end of the whole steps
  if (full_end - r0 >= 2 * X3_DEPTH * X3_ROWS) {
#pragma unroll
    for (int d = 0; d < X3_DEPTH; ++d) fetch_full(sa[d], sb[d]);
    for (; r + 2 * X3_DEPTH * X3_ROWS <= full_end; r += X3_DEPTH * X3_ROWS) {
#pragma unroll
      for (int d = 0; d < X3_DEPTH; ++d) {
        __syncthreads();  // the previous step's fragments have been read
        split_store(sa[d], sb[d]);
        fetch_full(sa[d], sb[d]);  // refill this set for step +DEPTH
        __syncthreads();
        multiply();
      }
    }
#pragma unroll
    for (int d = 0; d < X3_DEPTH; ++d) {  // drain the ring
      __syncthreads();
      split_store(sa[d], sb[d]);
      __syncthreads();
      multiply();
    }
    r += X3_DEPTH * X3_ROWS;
  }
  for (; r < r1; r += X3_ROWS) {  // what is left of the chunk (and chunks too short for the ring)
    fetch_any(sa[0], sb[0], r);
    __syncthreads();
    split_store(sa[0], sb[0]);
    __syncthreads();
    multiply();
  }
  flush();
  float* P = a.part + ((int64_t)blockIdx.x) * a.K * a.Nf;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = nbase + t * 32 + li;
    if (n < a.Nf) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = kbase + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        if (kk < a.K) P[(int64_t)kk * a.Nf + n] = sum[t][i];
      }
    }
  }
}

// A second form of this kernel was built and measured in round 2 and is NOT kept: one 512-thread
// block per CU, double-buffered plane images (one barrier per step), waves 0-3 splitting while waves
// 4-7 multiply (stagger), and a two-deep load ring in reserved registers (async_stage.h).  It ran
// 2.17 ms against 1.89 ms for the kernel above at R = 8 M rows (2.20 ms without the stagger,
// 2.02 ms without the RNE flush; profiles/archive/r02k_ab_dw_forms.txt).  The PMC pass of
// profiles/archive/r02l_pmc_sq_gemm_dw.json says why more prefetch does not help either form: the matrix
// pipes are busy 47-48 % of the cycles and the waves wait on instruction ISSUE (MFMA pipe / VALU)
// for 48 % of their life, on memory or barriers for only 26 %, at a shader clock of ~1.6 GHz — the
// six plane products plus ~200 VALU instructions of splitting per step are co-limiting with the
// 32.8 GB stream, they are not hidden under it.

// small dW: rows staged through LDS in tiles of DW_ROWS.  With few outputs (K*Nf <= 128, the
// reference's 2x6 / 6x6) the 256 threads form 256/(K*Nf) row groups that each take every g-th row
// of the tile, so all lanes work; with more outputs each thread owns up to 4 output elements.
// fp64 running sums (free at these sizes), groups combined through LDS in fixed order.
constexpr int DW_ROWS = 64;
__global__ __launch_bounds__(256) void gemm_dw_small_kernel(DwArgs a) {
  extern __shared__ float sm[];  // [DW_ROWS][K] then [DW_ROWS][Nf]
  __shared__ double red[256];
  float* sa = sm;
  float* sd = sm + DW_ROWS * a.K;
  const int64_t batch = blockIdx.x / a.chunks;
  const int chunk = blockIdx.x % a.chunks;
  const int64_t b0 = batch * a.batch_rows;
  int64_t b1 = b0 + a.batch_rows;
  if (b1 > a.R) b1 = a.R;
  const int64_t r0 = b0 + (int64_t)chunk * a.rows_per_chunk;
  int64_t r1 = r0 + a.rows_per_chunk;
  if (r1 > b1) r1 = b1;
  const int n_out = a.K * a.Nf;
  const int groups = n_out <= 128 ? 256 / n_out : 1;   // row groups (few outputs)
  const int grp = groups > 1 ? threadIdx.x / n_out : 0;
  constexpr int OMAX = 4;                                // output slots per thread (many outputs)
  double acc[OMAX];
  int ok_[OMAX], on_[OMAX];
#pragma unroll
  for (int o = 0; o < OMAX; ++o) {
    acc[o] = 0.0;
    int idx = groups > 1 ? (o == 0 && grp < groups ? (int)(threadIdx.x % n_out) : n_out) : threadIdx.x + o * 256;
    ok_[o] = idx < n_out ? idx / a.Nf : -1;
    on_[o] = idx < n_out ? idx % a.Nf : 0;
  }
  for (int64_t r = r0; r < r1; r += DW_ROWS) {
    const int nr = (int)((r1 - r) < DW_ROWS ? (r1 - r) : DW_ROWS);
    __syncthreads();
    for (int t = threadIdx.x; t < nr * a.K; t += 256) sa[t] = a.A[r * a.K + t];
    for (int t = threadIdx.x; t < nr * a.Nf; t += 256) sd[t] = a.dY[r * a.Nf + t];
    __syncthreads();
#pragma unroll
    for (int o = 0; o < OMAX; ++o) {
      if (ok_[o] >= 0) {
        double s = acc[o];
        for (int i = grp; i < nr; i += groups) s += (double)sa[i * a.K + ok_[o]] * (double)sd[i * a.Nf + on_[o]];
        acc[o] = s;
      }
    }
  }
  float* P = a.part + ((int64_t)blockIdx.x) * n_out;
  if (groups > 1) {
    __syncthreads();
    red[threadIdx.x] = acc[0];
    __syncthreads();
    if (threadIdx.x < n_out) {
      double s = 0.0;
      for (int g = 0; g < groups; ++g) s += red[g * n_out + threadIdx.x];
      P[threadIdx.x] = (float)s;
    }
  } else {
#pragma unroll
    for (int o = 0; o < OMAX; ++o)
      if (ok_[o] >= 0) P[threadIdx.x + o * 256] = (float)acc[o];
  }
}

// Narrow layers (K, Nf even and <= 8: the scripts' 2x6, 6x6, 6x2), both widths known at compile time.
// FOUR lanes share a row: lane j of a group takes the (K/2) x (Nf/2) block (k-half j >> 1, n-half j & 1) of the row's
// outer product — it loads only its halves of the A and dY rows (the four lanes together read each row once, whole
// cache lines) and keeps K·Nf/4 fp64 sums instead of K·Nf (9 instead of 36 for the 6x6 weight: 18 VGPRs, not 72), which
// leaves room for FOUR rows in flight per lane (the one-row form was bound by its load -> fma dependency, and four rows
// next to 72 accumulator registers were slower: round 2).  The block folds once: xor butterfly over the lane bits above
// the group (4 stages of K·Nf/4 values, not 6 of K·Nf), then the four waves in order.
// ACT: dY is multiplied by act'(pre) on the fly (dW = Aᵀ·(dY ⊙ act'(pre)), autograd of act(A·W) with respect to W) — the
// layer-1 backward of the 2-layer models without the [T,N,F] act_bwd pass in between.
template <int KT, int NT, bool ACT>
__global__ __launch_bounds__(256) void gemm_dw_narrow_kernel(DwArgs a) {
  constexpr int NO = KT * NT, KH = KT / 2, NH = NT / 2;
  __shared__ double red[4][NO];
  const int64_t batch = blockIdx.x / a.chunks;
  const int chunk = blockIdx.x % a.chunks;
  const int64_t b0 = batch * a.batch_rows;
  int64_t b1 = b0 + a.batch_rows;
  if (b1 > a.R) b1 = a.R;
  const int64_t r0 = b0 + (int64_t)chunk * a.rows_per_chunk;
  int64_t r1 = r0 + a.rows_per_chunk;
  if (r1 > b1) r1 = b1;
  const int j = threadIdx.x & 3, kh = j >> 1, nh = j & 1;
  double acc[KH][NH];
#pragma unroll
  for (int k = 0; k < KH; ++k)
#pragma unroll
    for (int n = 0; n < NH; ++n) acc[k][n] = 0.0;
  // RUN rows in flight per lane (rows r, r + 64, …: a block covers 64·RUN rows per trip).  No bounds test inside: a load
  // under a per-lane condition compiles to a branch with a full vmcnt(0) wait per element, so full trips run
  // unconditionally and the last rows of the chunk take single-row trips.
  const ActGrad dact(a.act);
  auto trip = [&](int64_t r, auto run_tag) {
    constexpr int RUN = decltype(run_tag)::value;
    float x[RUN][KH], g[RUN][NH], pa[RUN][ACT ? NH : 1];
#pragma unroll
    for (int u = 0; u < RUN; ++u) {
      const float* xa = a.A + (r + 64 * u) * KT + kh * KH;
      const float* ga = a.dY + (r + 64 * u) * NT + nh * NH;
#pragma unroll
      for (int i = 0; i < KH; ++i) x[u][i] = xa[i];
#pragma unroll
      for (int i = 0; i < NH; ++i) g[u][i] = ga[i];
      if constexpr (ACT) {
        const float* pp = a.pre + (r + 64 * u) * NT + nh * NH;
#pragma unroll
        for (int i = 0; i < NH; ++i) pa[u][i] = pp[i];
      }
    }
    if constexpr (ACT) {
#pragma unroll
      for (int u = 0; u < RUN; ++u)
#pragma unroll
        for (int i = 0; i < NH; ++i) g[u][i] *= dact(pa[u][i]);
    }
#pragma unroll
    for (int u = 0; u < RUN; ++u)
#pragma unroll
      for (int k = 0; k < KH; ++k)
#pragma unroll
        for (int n = 0; n < NH; ++n) acc[k][n] = fma((double)x[u][k], (double)g[u][n], acc[k][n]);
  };
  constexpr int RU = 4;
  int64_t r = r0 + (threadIdx.x >> 2);
  for (; r + 64 * (RU - 1) < r1; r += 64 * RU) trip(r, std::integral_constant<int, RU>{});
  for (; r < r1; r += 64) trip(r, std::integral_constant<int, 1>{});
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KH; ++k)
#pragma unroll
    for (int n = 0; n < NH; ++n) {
      double v = acc[k][n];
#pragma unroll
      for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o);
      if (lane < 4) red[wave][(kh * KH + k) * NT + nh * NH + n] = v;
    }
  __syncthreads();
  // slab stored write-through (4-byte relaxed agent-scope atomic store); the last block to finish adds the slabs
  // of every batch in a fixed order and writes dW — no second launch (common.h: last_block_ticket)
  if (threadIdx.x < NO)
    __hip_atomic_store(reinterpret_cast<unsigned*>(a.part) + (int64_t)blockIdx.x * NO + threadIdx.x,
                       __float_as_uint((float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x])),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.n_batch == 1) {
    // one output matrix: the slabs are added as a tree over the two ticket levels (common.h: slab_tree_finish)
    __shared__ double total[NO];
    if (!slab_tree_finish<NO>(reinterpret_cast<unsigned*>(a.part), (int)gridDim.x, a.sync, total)) return;
    if (threadIdx.x < NO) a.dW[threadIdx.x] = (float)total[threadIdx.x];
  } else {
    // one weight per slice: few chunks per batch, many outputs — the last block, a thread per output walking its chunks in order
    __shared__ int is_last;
    if (!last_block_ticket(a.sync, (int)gridDim.x, &is_last)) return;
    if (threadIdx.x == 0) *a.sync = 0;
    const unsigned* P = reinterpret_cast<const unsigned*>(a.part);
    const int64_t total = (int64_t)a.n_batch * NO;
    for (int64_t idx = threadIdx.x; idx < total; idx += 256) {
      const int64_t b = idx / NO, o = idx - b * NO;
      double sum = 0.0;
      for (int c = 0; c < a.chunks; ++c)
        sum += (double)__uint_as_float(__hip_atomic_load(P + (b * a.chunks + c) * NO + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      a.dW[idx] = (float)sum;
    }
  }
}

// dW[b][o] = sum over chunks of part[b][chunk][o].  A block owns 64 consecutive outputs; its sixteen
// waves take the chunks c = w, w+16, ... (lanes along the outputs: every load is one coalesced 256-byte
// row of a slab), four independent fp64 sums per lane in flight, combined in a fixed order.  (Four
// waves per block walked the slabs as 96 rounds of dependent-latency loads: 106 us for 1250 slabs of
// 128x128, 211 us of a 4.1 ms wide-feature epoch.)
constexpr int DWR_WAVES = 16;
__global__ __launch_bounds__(DWR_WAVES * 64) void gemm_dw_reduce_kernel(const float* __restrict__ part,
                                                                         float* __restrict__ dW, int64_t n_out,
                                                                         int32_t chunks, int64_t total) {
  __shared__ double sh[DWR_WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t idx = (int64_t)blockIdx.x * 64 + lane;       // flat (batch, output)
  const bool in = idx < total;
  const int64_t b = in ? idx / n_out : 0, o = in ? idx % n_out : 0;
  const float* p = part + b * chunks * n_out + o;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int c = w;
  for (; c + 3 * DWR_WAVES < chunks; c += 4 * DWR_WAVES) {
    const float v0 = in ? p[(int64_t)c * n_out] : 0.f, v1 = in ? p[(int64_t)(c + DWR_WAVES) * n_out] : 0.f;
    const float v2 = in ? p[(int64_t)(c + 2 * DWR_WAVES) * n_out] : 0.f,
                v3 = in ? p[(int64_t)(c + 3 * DWR_WAVES) * n_out] : 0.f;
    s0 += (double)v0;
    s1 += (double)v1;
    s2 += (double)v2;
    s3 += (double)v3;
  }
  for (; c < chunks; c += DWR_WAVES) s0 += in ? (double)p[(int64_t)c * n_out] : 0.0;
  sh[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && in) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < DWR_WAVES; ++i) t += sh[i][lane];
    dW[idx] = (float)t;
  }
}

// The same reduction for narrow layers (a handful of outputs, hundreds of slabs — the scripts' 2x6
// and 6x2 weights): one block per output, its 256 threads stride over the slabs, xor butterfly per
// wave, the four waves in order.  The kernel above would walk the slabs as one serial chain of
// dependent-latency loads per wave (30 us for 12 outputs x 483 slabs, measured).
__global__ __launch_bounds__(256) void gemm_dw_reduce_narrow_kernel(const float* __restrict__ part,
                                                                     float* __restrict__ dW, int64_t n_out,
                                                                     int32_t chunks) {
  __shared__ double sh[4];
  const int64_t idx = blockIdx.x;  // flat (batch, output)
  const int64_t b = idx / n_out, o = idx % n_out;
  const float* p = part + b * chunks * n_out + o;
  double s = 0.0;
  for (int c = threadIdx.x; c < chunks; c += 256) s += (double)p[(int64_t)c * n_out];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dW[idx] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

static bool use_small(int K, int Nf) { return (K < 16 || Nf < 16) && K <= 64 && Nf <= 64; }

static void dw_plan(int64_t R, int64_t rows_per_batch, int64_t* n_batch, int* chunks,
                    int64_t* rows_per_chunk) {
  const int64_t br = rows_per_batch ? rows_per_batch : R;
  const int64_t nb = br ? (R + br - 1) / br : 1;
  int64_t c = (br + 511) / 512;  // >= 512 rows per chunk
  int64_t cmax = 1536 / (nb > 0 ? nb : 1);  // three rounds of the 512 blocks (2 per CU) the bf16x3 kernel keeps resident
  if (cmax < 1) cmax = 1;
  if (c > cmax) c = cmax;
  if (c < 1) c = 1;
  int64_t rpc = (br + c - 1) / c;
  rpc = (rpc + 31) & ~(int64_t)31;  // whole unrolled steps (2 * UR rows, UR <= 16)
  c = (br + rpc - 1) / (rpc ? rpc : 1);
  if (c < 1) c = 1;
  *n_batch = nb;
  *chunks = (int)c;
  *rows_per_chunk = rpc;
}

}  // namespace tmgcn

using namespace tmgcn;

static int gemm_launch(const float* A, const void* W, bool w_bf16, float* Y, float* pre_act, int64_t R,
                       int32_t K, int32_t Nf, int32_t trans_w, int64_t rows_per_batch,
                       int64_t w_batch_stride, int32_t act, int32_t algo, void* stream) {
  TMGCN_REQUIRE(R >= 0 && K > 0 && Nf > 0, "gemm: bad shape R=%lld K=%d Nf=%d", (long long)R, K, Nf);
  TMGCN_REQUIRE(algo == TMGCN_GEMM_AUTO || algo == TMGCN_GEMM_F32MFMA, "gemm: unknown algo %d", algo);
  TMGCN_REQUIRE(rows_per_batch >= 0, "gemm: negative rows_per_batch");
  TMGCN_REQUIRE(act >= TMGCN_ACT_NONE && act <= TMGCN_ACT_SELU, "gemm: unknown activation %d", act);
  if (R == 0) return TMGCN_OK;
  TMGCN_REQUIRE(A && W && Y, "gemm: null pointer");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{A, static_cast<const float*>(W), Y, pre_act, R, K, Nf, trans_w, rows_per_batch, w_batch_stride,
             act, 0, 0, nullptr, w_bf16 ? 1 : 0, 0, 0, 0, 0};
  if (use_small(K, Nf)) {
    const int Nfp = (Nf + 7) & ~7;
    const int64_t br = rows_per_batch ? rows_per_batch : R;
    const int64_t max_w = br >= 256 ? 2 : (256 / br + 2);
    size_t smem = (size_t)max_w * K * Nfp * sizeof(float);
    TMGCN_REQUIRE(smem <= 64 * 1024, "gemm: per-slice weights too small a batch (%lld rows)",
                  (long long)br);
    if (Nfp <= 32 && smem + (size_t)256 * (Nfp + 1) * sizeof(float) <= 64 * 1024) {  // output tile through LDS: 256 rows x (Nfp + 1) floats, <= 33 KB; the launch stays within 64 KB of dynamic LDS
      a.stage_off = (int32_t)(smem / sizeof(float));
      smem += (size_t)256 * (Nfp + 1) * sizeof(float);
    }
    const unsigned grid = (unsigned)((R + 255) / 256);
    hipLaunchKernelGGL(gemm_small_kernel, dim3(grid), dim3(256), smem, st, a);
    return check_launch("gemm_small");
  }
  const int64_t br = rows_per_batch ? rows_per_batch : R;
  const int64_t nb = (R + br - 1) / br;
  a.tiles_per_batch = (br + BM - 1) / BM;
  a.n_tiles = nb * a.tiles_per_batch;
  const unsigned gy = (unsigned)((Nf + 127) / 128);
  TMGCN_REQUIRE(a.n_tiles < (int64_t)0x7fffffff && gy <= 64, "gemm: shape too large for the tile scheduler");
  const bool x3 = algo == TMGCN_GEMM_AUTO && K % 4 == 0 && K >= 16 && K <= 128 &&
                  reinterpret_cast<uintptr_t>(A) % 16 == 0;
  // 128 < K <= 512: the same kernel once per 128-wide k-chunk, every chunk after the first adding to
  // Y (one more read of Y per chunk; still ~1.5-1.8x the exact-f32 MFMA kernel, which is compute-bound)
  const bool x3_chunked = algo == TMGCN_GEMM_AUTO && K % 4 == 0 && K > 128 && K <= 512 &&
                          reinterpret_cast<uintptr_t>(A) % 16 == 0;
  if (x3_chunked) {
    int64_t gx = w_bf16 ? persistent_grid(gemm_bf16x3_kernel<1, true>, 256) : persistent_grid(gemm_bf16x3_kernel<3, true>, 256);
    if (gx > a.n_tiles) gx = a.n_tiles;
    for (int k0 = 0; k0 < K; k0 += 128) {
      GemmArgs c = a;
      const bool last = k0 + 128 >= K;
      c.K = last ? K - k0 : 128;
      c.lda = K;
      c.k0 = k0;
      c.accum = k0 > 0;
      c.act = last ? act : TMGCN_ACT_NONE;
      c.pre = last ? pre_act : nullptr;
      if (w_bf16)
        hipLaunchKernelGGL((gemm_bf16x3_kernel<1, true>), dim3((unsigned)gx, gy), dim3(256), 0, st, c);
      else
        hipLaunchKernelGGL((gemm_bf16x3_kernel<3, true>), dim3((unsigned)gx, gy), dim3(256), 0, st, c);
    }
    return check_launch("gemm_bf16x3 (k-chunks)");
  }
  if (x3) {  // static persistent schedule: no tile counter
    int64_t gx = w_bf16 ? persistent_grid(gemm_bf16x3_kernel<1>, 256) : persistent_grid(gemm_bf16x3_kernel<3>, 256);
    if (gx > a.n_tiles) gx = a.n_tiles;
    if (w_bf16)
      hipLaunchKernelGGL(gemm_bf16x3_kernel<1>, dim3((unsigned)gx, gy), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(gemm_bf16x3_kernel<3>, dim3((unsigned)gx, gy), dim3(256), 0, st, a);
    return check_launch("gemm_bf16x3");
  }
  // gy consecutive counters of the pool (acquire zeroes one; take gy of them in a row)
  a.tile_counter = acquire_tile_counters(st, (int)gy);
  TMGCN_REQUIRE(a.tile_counter, "gemm: no tile counters: %s", pool_error());
  int64_t gx = persistent_grid(gemm_mfma_kernel, 256);
  if (gx > a.n_tiles) gx = a.n_tiles;
  hipLaunchKernelGGL(gemm_mfma_kernel, dim3((unsigned)gx, gy), dim3(256), 0, st, a);
  return check_launch("gemm_mfma");
}

extern "C" int tmgcn_gemm_f32(const float* A, const float* W, float* Y, float* pre_act, int64_t R,
                               int32_t K, int32_t Nf, int32_t trans_w, int64_t rows_per_batch,
                               int64_t w_batch_stride, int32_t act, int32_t algo, void* stream) {
  return gemm_launch(A, W, false, Y, pre_act, R, K, Nf, trans_w, rows_per_batch, w_batch_stride, act, algo, stream);
}

extern "C" int tmgcn_gemm_bf16w_f32(const float* A, const uint16_t* W_bf16, float* Y, float* pre_act, int64_t R,
                                     int32_t K, int32_t Nf, int32_t trans_w, int64_t rows_per_batch,
                                     int64_t w_batch_stride, int32_t act, int32_t algo, void* stream) {
  return gemm_launch(A, W_bf16, true, Y, pre_act, R, K, Nf, trans_w, rows_per_batch, w_batch_stride, act, algo,
                     stream);
}

extern "C" int64_t tmgcn_gemm_dw_workspace_bytes(int64_t R, int32_t K, int32_t Nf,
                                                  int64_t rows_per_batch) {
  if (R <= 0 || K <= 0 || Nf <= 0) return 0;
  int64_t nb, rpc;
  int chunks;
  dw_plan(R, rows_per_batch, &nb, &chunks, &rpc);
  return (nb * chunks + kSyncGroups) * (int64_t)K * Nf * (int64_t)sizeof(float);      // block slabs + the group slabs of slab_tree_finish
}

static int gemm_dw_launch(const float* A, const float* dY, const float* pre, int32_t act, float* dW, int64_t R, int32_t K,
                          int32_t Nf, int64_t rows_per_batch, int32_t algo, void* workspace, int64_t workspace_bytes,
                          void* stream) {
  TMGCN_REQUIRE(R >= 0 && K > 0 && Nf > 0, "gemm_dw: bad shape");
  TMGCN_REQUIRE(algo == TMGCN_DW_AUTO || algo == TMGCN_DW_F32MFMA, "gemm_dw: unknown algo %d", algo);
  TMGCN_REQUIRE(rows_per_batch >= 0, "gemm_dw: negative rows_per_batch");
  TMGCN_REQUIRE(dW, "gemm_dw: null dW");
  hipStream_t st = (hipStream_t)stream;
  int64_t nb, rpc;
  int chunks;
  dw_plan(R, rows_per_batch, &nb, &chunks, &rpc);
  if (R == 0) {
    (void)hipMemsetAsync(dW, 0, (size_t)K * Nf * sizeof(float), st);
    return check_launch("gemm_dw memset");
  }
  TMGCN_REQUIRE(A && dY, "gemm_dw: null pointer");
  const int64_t need = (nb * chunks + kSyncGroups) * (int64_t)K * Nf * (int64_t)sizeof(float);
  if (!workspace || workspace_bytes < need) {
    set_error("gemm_dw: workspace %lld B < required %lld B", (long long)workspace_bytes,
              (long long)need);
    return TMGCN_ERR_WORKSPACE;
  }
  const bool narrow = K % 2 == 0 && Nf % 2 == 0 && K <= 8 && Nf <= 8 && reinterpret_cast<uintptr_t>(A) % 8 == 0 &&
                      reinterpret_cast<uintptr_t>(dY) % 8 == 0;
  TMGCN_REQUIRE(!pre || narrow, "gemm_dw_act: the fused activation gradient needs the narrow kernel (even K, Nf <= 8, 8-byte "
                                 "aligned operands); use tmgcn_act_bwd_f32 + tmgcn_gemm_dw_f32");
  if (narrow) {
    // The narrow kernel's fixed costs are per block (the K·Nf-value block reduction, the slab and the hand-off tickets),
    // its loop is a stream: about two blocks per CU while the operand is small (captured steps, kernel durations under
    // rocprofv3, slabs reduced as a tree: 570 k rows 9.8 us with 509 blocks, 11.6 with 279 or 1 018; 150 k rows 7.6 - 8.0 us
    // from 147 to 469 blocks), four per CU from 4 M rows on; never more slabs than planned (workspace).
    const int64_t br = rows_per_batch ? rows_per_batch : R;
    int64_t target = (R < (4ll << 20) ? 512 : 1024) / nb;      // blocks per batch
    if (target < 1) target = 1;
    int64_t fat = (br + target - 1) / target;
    fat = (fat + 31) & ~(int64_t)31;
    if (fat < 256) fat = 256;
    if (fat > rpc) {
      rpc = fat;
      chunks = (int)((br + rpc - 1) / rpc);
    }
  }
  DwArgs a{A, dY, (float*)workspace, R, K, Nf, rows_per_batch ? rows_per_batch : R, chunks, rpc};
  const unsigned gx = (unsigned)(nb * chunks);
  if (narrow) {
    a.dW = dW;
    a.pre = pre;
    a.act = act;
    a.n_batch = (int32_t)nb;
    a.sync = acquire_sync_word(st);
    TMGCN_REQUIRE(a.sync, "gemm_dw: no hand-off block: %s", pool_error());
#define TMGCN_DWN_L(KT_, NT_)                                                                          \
  if (pre) hipLaunchKernelGGL((gemm_dw_narrow_kernel<KT_, NT_, true>), dim3(gx), dim3(256), 0, st, a); \
  else hipLaunchKernelGGL((gemm_dw_narrow_kernel<KT_, NT_, false>), dim3(gx), dim3(256), 0, st, a);
#define TMGCN_DWN_N(KT_)                \
  switch (Nf) {                         \
    case 2: TMGCN_DWN_L(KT_, 2) break;  \
    case 4: TMGCN_DWN_L(KT_, 4) break;  \
    case 6: TMGCN_DWN_L(KT_, 6) break;  \
    default: TMGCN_DWN_L(KT_, 8)        \
  }
    switch (K) {
      case 2: TMGCN_DWN_N(2) break;
      case 4: TMGCN_DWN_N(4) break;
      case 6: TMGCN_DWN_N(6) break;
      default: TMGCN_DWN_N(8)
    }
#undef TMGCN_DWN_N
#undef TMGCN_DWN_L
    return check_launch("gemm_dw_narrow");          // reduced by its own last block
  } else if (use_small(K, Nf)) {
    const size_t smem = (size_t)DW_ROWS * (K + Nf) * sizeof(float);
    hipLaunchKernelGGL(gemm_dw_small_kernel, dim3(gx), dim3(256), smem, st, a);
  } else {
    const unsigned gy = (unsigned)(((K + 127) / 128) * ((Nf + 127) / 128));
    const bool x3 = algo == TMGCN_DW_AUTO && K % 4 == 0 && Nf % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(dY) % 16 == 0;
    if (x3)
      hipLaunchKernelGGL(gemm_dw_bf16x3_kernel, dim3(gx, gy), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(gemm_dw_lds_kernel, dim3(gx, gy), dim3(256), 0, st, a);
  }
  int rc = check_launch("gemm_dw");
  if (rc) return rc;
  const int64_t n_out = (int64_t)K * Nf;
  const int64_t total = nb * n_out;
  if (total <= 1024 && chunks >= 64)
    hipLaunchKernelGGL(gemm_dw_reduce_narrow_kernel, dim3((unsigned)total), dim3(256), 0, st, (const float*)workspace,
                       dW, n_out, chunks);
  else
    hipLaunchKernelGGL(gemm_dw_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(DWR_WAVES * 64), 0, st,
                       (const float*)workspace, dW, n_out, chunks, total);
  return check_launch("gemm_dw_reduce");
}

extern "C" int tmgcn_gemm_dw_f32(const float* A, const float* dY, float* dW, int64_t R, int32_t K, int32_t Nf,
                                  int64_t rows_per_batch, int32_t algo, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  return gemm_dw_launch(A, dY, nullptr, TMGCN_ACT_NONE, dW, R, K, Nf, rows_per_batch, algo, workspace, workspace_bytes, stream);
}

extern "C" int tmgcn_gemm_dw_act_supported(int32_t K, int32_t Nf) {
  return (K >= 2 && Nf >= 2 && K % 2 == 0 && Nf % 2 == 0 && K <= 8 && Nf <= 8) ? 1 : 0;
}

extern "C" int tmgcn_gemm_dw_act_f32(const float* A, const float* dY, const float* pre_act, int32_t act, float* dW,
                                      int64_t R, int32_t K, int32_t Nf, int64_t rows_per_batch, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  TMGCN_REQUIRE(tmgcn_gemm_dw_act_supported(K, Nf), "gemm_dw_act: even K, Nf <= 8 only (got %d x %d)", K, Nf);
  TMGCN_REQUIRE(act >= TMGCN_ACT_NONE && act <= TMGCN_ACT_SELU, "gemm_dw_act: unknown activation %d", act);
  TMGCN_REQUIRE(pre_act && reinterpret_cast<uintptr_t>(pre_act) % 8 == 0, "gemm_dw_act: pre_act must be given, 8-byte aligned");
  return gemm_dw_launch(A, dY, pre_act, act, dW, R, K, Nf, rows_per_batch, TMGCN_DW_AUTO, workspace, workspace_bytes, stream);
}

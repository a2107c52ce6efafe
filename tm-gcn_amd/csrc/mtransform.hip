// P1 — tube-fibre M-transform  Y[k][c] = sum_j Mop[ro+k][co+j] * X[j][c]   (gfx950 / CDNA4)
//
// Replaces  t.matmul(self.M, X.reshape(self.T,-1)).reshape(X.size())
// (embedding_help_functions.py:204, 308, 346, 404), the Minv product (ehf:224, 332, 341)
// and autograd's Mᵀ product.  X is viewed as [T_in][C], C = N*F: every load and store is
// a 16-B-per-lane access along C (fully coalesced); the T×T operator is tiny and is read
// through the scalar cache (its index is wave-uniform).
//
// Two kernels:
//   mtransform_band<W>  the operator is banded (all of the reference's M are lower-banded
//                       with <= 20 diagonals, read_data.m:116-124, SBM_our.py:88-96; Mᵀ is
//                       upper-banded).  Each lane keeps a W-row sliding window of its four
//                       columns in registers, so X is read once and Y written once:
//                       8 B/element of HBM traffic, 2W flop/element -> HBM-bound.
//   mtransform_bf16x3   any operator with T_in <= 128 (Minv, bands wider than 20): bf16 matrix cores after
//                       an exact 3-way split of both operands (fp32-accurate), a stream over X and Y.
//   mtransform_dense_mfma / mtransform_dense   exact-f32 MFMA (T_in <= 256) / register-blocked FMA fallbacks.
#include <type_traits>
#include "common.h"
#include "async_stage.h"

namespace tmgcn {

struct MtArgs {
  const float* M;
  int32_t ldm;
  int32_t transpose;
  int32_t row_off, col_off;
  int32_t T_out, T_in;
  int32_t band_lo, band_hi;
  const float* X;
  float* Y;
  int64_t C;  // columns (floats)
  int32_t rows_per_chunk;
  int32_t x_tl, y_tl;  // group-interleaved row storage (0 = plain row order), see tmgcn.h
  unsigned int* tile_counter;  // dense MFMA kernel: dynamic column-tile scheduling (common.h)
  int64_t ldx, ldy;            // row strides of X and Y in floats (>= C): a column window of a wider tensor
};

// storage position of logical row k of a tensor with T rows stored in groups of tl rows:
// (k % tl) * (T / tl) + k / tl — the send/receive layout of the slice<->node all-to-all.
__device__ __forceinline__ int64_t row_pos(int k, int T, int tl) {
  return tl ? (int64_t)(k % tl) * (T / tl) + k / tl : k;
}

__device__ __forceinline__ float mop(const MtArgs& a, int k, int j) {
  const int64_t r = a.row_off + k, c = a.col_off + j;
  return a.transpose ? a.M[c * a.ldm + r] : a.M[r * a.ldm + c];
}

// VEC-wide column access helpers (VEC = 4: float4, VEC = 1: scalar tail / unaligned)
template <int VEC>
struct Cols;
template <>
struct Cols<4> {
  using T = float4;
  static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ T load(const float* p) {
    return *reinterpret_cast<const float4*>(p);
  }
  static __device__ __forceinline__ void store(float* p, const T& v) {
    *reinterpret_cast<float4*>(p) = v;
  }
  static __device__ __forceinline__ void fma(T& acc, float m, const T& x) {
    acc.x = fmaf(m, x.x, acc.x);
    acc.y = fmaf(m, x.y, acc.y);
    acc.z = fmaf(m, x.z, acc.z);
    acc.w = fmaf(m, x.w, acc.w);
  }
  static __device__ __forceinline__ void mask(T& v, unsigned bits) {
    v.x = __uint_as_float(__float_as_uint(v.x) & bits);
    v.y = __uint_as_float(__float_as_uint(v.y) & bits);
    v.z = __uint_as_float(__float_as_uint(v.z) & bits);
    v.w = __uint_as_float(__float_as_uint(v.w) & bits);
  }
};
template <>
struct Cols<1> {
  using T = float;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, const T& v) { *p = v; }
  static __device__ __forceinline__ void fma(T& acc, float m, const T& x) { acc = fmaf(m, x, acc); }
  static __device__ __forceinline__ void mask(T& v, unsigned bits) {
    v = __uint_as_float(__float_as_uint(v) & bits);
  }
};

// Sliding-window band kernel.  Each lane owns VEC columns and keeps a ring of W = WIDTH + PF
// input rows in registers: WIDTH rows feed the current output row while the next PF rows are
// already in flight, so X is read once, Y written once, and every lane keeps PF 16-B loads
// outstanding.  Rows are numbered locally, u = 0.. (u = 0 is the oldest row the chunk's first
// output needs); row u lives in ring slot u mod W.  Three phases keep the steady state free of
// control flow (conditional loads would force s_waitcnt vmcnt(0) and serialise the stream):
//   fill    rows 0..WIDTH-2 and the first PF look-ahead rows: loads only, fully unrolled
//   steady  groups of W output rows, fully unrolled, unconditional loads (row index clamped
//           into the tensor, out-of-range rows selected to zero), one store per row
//   tail    the last < W rows, same body under wave-uniform branches
template <int WIDTH, int PF, int VEC, bool PERM>
struct BandBody {
  static constexpr int W = WIDTH + PF;
  using CT = Cols<VEC>;
  using V = typename CT::T;

  // input row of local index u
  static __device__ __forceinline__ V fetch(const MtArgs& a, const float* __restrict__ X, int q0,
                                            int u, int u_max, int64_t c) {
    const int uc = u < u_max ? u : u_max;  // never run past the rows this chunk needs
    const int q = q0 + uc;
    const int qc = q < 0 ? 0 : (q >= a.T_in ? a.T_in - 1 : q);
    const int64_t pos = PERM ? row_pos(qc, a.T_in, a.x_tl) : (int64_t)qc;
    V v = CT::load(X + pos * a.ldx + c);
    CT::mask(v, q == qc ? 0xFFFFFFFFu : 0u);  // rows outside the tensor are zero (bit mask: no branch)
    return v;
  }

  // output row o of the chunk, completed by the row in ring slot i; taps from the LDS table
  static __device__ __forceinline__ void emit(const MtArgs& a, float* __restrict__ Y, const V (&win)[W],
                                              const float* coef, int i, int o, int k, int64_t c, bool live) {
    V acc = CT::zero();
    float m[WIDTH];
#pragma unroll
    for (int d = 0; d < WIDTH; ++d) m[d] = coef[o * WIDTH + d];  // wave-uniform: LDS broadcast
#pragma unroll
    for (int d = 0; d < WIDTH; ++d) CT::fma(acc, m[d], win[(i - d + 2 * W) % W]);
    const int64_t pos = PERM ? row_pos(k, a.T_out, a.y_tl) : (int64_t)k;
    if (live) CT::store(Y + pos * a.ldy + c, acc);
  }
};

constexpr int kBandMaxChunkRows = 512;  // LDS tap table: 512 x 20 x 4 B = 40 KB

template <int WIDTH, int PF, int VEC, bool PERM>
__global__ __launch_bounds__(256) void mtransform_band_kernel(MtArgs a) {
  using B = BandBody<WIDTH, PF, VEC, PERM>;
  using CT = Cols<VEC>;
  using V = typename CT::T;
  constexpr int W = B::W;
  extern __shared__ float coef[];  // [n_out][WIDTH] taps of this chunk, zero outside the band
  const int k_begin = blockIdx.y * a.rows_per_chunk;
  int k_end = k_begin + a.rows_per_chunk;
  if (k_end > a.T_out) k_end = a.T_out;
  const int n_out = k_end - k_begin;
  if (n_out <= 0) return;  // whole block
  int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  const bool live = c < a.C;
  if (!live) c = 0;  // idle lanes of the last block shadow column 0 and never store
  const float* __restrict__ X = a.X;
  float* __restrict__ Y = a.Y;

  // output row k reads input rows [k + d_lo, k + d_hi]; tap d of row k is input row k + d_hi - d
  const int d_lo = (a.row_off - a.col_off) - a.band_lo;
  const int d_hi = (a.row_off - a.col_off) + a.band_hi;
  for (int t = threadIdx.x; t < n_out * WIDTH; t += blockDim.x) {
    const int o = t / WIDTH, d = t % WIDTH;
    const int k = k_begin + o, j = k + d_hi - d;
    coef[t] = (j >= k + d_lo && j >= 0 && j < a.T_in) ? mop(a, k, j) : 0.f;
  }
  __syncthreads();

  const int q0 = k_begin + d_hi - (WIDTH - 1);  // input row of local index 0
  const int u_max = n_out + WIDTH - 2;          // newest local row any output needs

  V win[W];
  // ---- fill: local rows 0 .. WIDTH-2+PF go to slots 0 .. W-2
#pragma unroll
  for (int u = 0; u < WIDTH - 1 + PF; ++u) win[u] = B::fetch(a, X, q0, u, u_max, c);

  // ---- steady: output o is completed by local row u = WIDTH-1+o in slot (WIDTH-1+o) % W.
  //      Row u + PF is fetched first, into slot (u + PF) % W = (u - WIDTH) % W: the row that
  //      lived there is older than anything output o (or any later one) reads.
  int o = 0;
  for (; o + W <= n_out; o += W) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const int u = WIDTH - 1 + o + i;
      win[(WIDTH - 1 + i + PF) % W] = B::fetch(a, X, q0, u + PF, u_max, c);
      B::emit(a, Y, win, coef, (WIDTH - 1 + i) % W, o + i, k_begin + o + i, c, live);
    }
  }
  // ---- tail: fewer than W outputs left (wave-uniform branches)
#pragma unroll
  for (int i = 0; i < W - 1; ++i) {
    if (o + i < n_out) {
      const int u = WIDTH - 1 + o + i;
      if (u + PF <= u_max) win[(WIDTH - 1 + i + PF) % W] = B::fetch(a, X, q0, u + PF, u_max, c);
      B::emit(a, Y, win, coef, (WIDTH - 1 + i) % W, o + i, k_begin + o + i, c, live);
    }
  }
}

// Dense fallback: each wave owns RT output rows, each lane VEC columns.
template <int RT, int VEC>
__global__ __launch_bounds__(256) void mtransform_dense_kernel(MtArgs a) {
  using CT = Cols<VEC>;
  using V = typename CT::T;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t c = ((int64_t)blockIdx.x * kWave + lane) * VEC;
  const int k0 = (blockIdx.y * 4 + wave) * RT;
  if (c >= a.C || k0 >= a.T_out) return;
  V acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) acc[i] = CT::zero();
  const int d_lo = (a.row_off - a.col_off) - a.band_lo;
  const int d_hi = (a.row_off - a.col_off) + a.band_hi;
  int j_lo = k0 + d_lo;
  if (j_lo < 0) j_lo = 0;
  int j_hi = k0 + RT - 1 + d_hi;
  if (j_hi > a.T_in - 1) j_hi = a.T_in - 1;
  for (int j = j_lo; j <= j_hi; ++j) {
    const V x = CT::load(a.X + row_pos(j, a.T_in, a.x_tl) * a.ldx + c);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int k = k0 + i;
      if (k < a.T_out && j >= k + d_lo && j <= k + d_hi) CT::fma(acc[i], mop(a, k, j), x);
    }
  }
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int k = k0 + i;
    if (k < a.T_out) CT::store(a.Y + row_pos(k, a.T_out, a.y_tl) * a.ldy + c, acc[i]);
  }
}

// Dense operator on the matrix cores (Minv = the reference's use_Minv path, ehf:224/332/341, or
// any M wider than 20 diagonals).  Y[T_out x C] = Mop · X is a GEMM whose long dimension is C:
//   A operand = Mop: wave w of a block owns output rows [32(4y+w), +32); its A fragments
//               (T_in/2 values per lane) stay in registers for the whole launch
//   B operand = a 64-column tile of X staged once per block through LDS (full-line loads),
//               read back as ds_read_b32 (lane = column: conflict-free)
//   v_mfma_f32_32x32x2_f32, exact fp32; k-steps outside the band of the wave's rows are skipped
//   (a lower-triangular Minv costs half the MFMAs).  T_in <= 2*SMAX.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kDenseCols = 64;

template <int SMAX>
__global__ __launch_bounds__(256) void mtransform_dense_mfma_kernel(MtArgs a) {
  extern __shared__ float Xs[];  // [T_in][64]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int k0 = (blockIdx.y * 4 + wave) * 32;
  const bool wave_live = k0 < a.T_out;
  const int d_lo = (a.row_off - a.col_off) - a.band_lo;
  const int d_hi = (a.row_off - a.col_off) + a.band_hi;
  const int nsteps = (a.T_in + 1) / 2;

  // A fragments: lane (i, h) holds Mop[k0+i][2s+h]
  float areg[SMAX];
#pragma unroll
  for (int s = 0; s < SMAX; ++s) {
    const int k = k0 + li, j = 2 * s + lh;
    float m = 0.f;
    if (s < nsteps && k < a.T_out && j < a.T_in && j >= k + d_lo && j <= k + d_hi) m = mop(a, k, j);
    areg[s] = m;
  }
  // k-steps this wave's 32 rows can touch
  int j_lo = k0 + d_lo, j_hi = k0 + 31 + d_hi;
  if (j_lo < 0) j_lo = 0;
  if (j_hi > a.T_in - 1) j_hi = a.T_in - 1;
  const int s_lo = j_lo / 2, s_hi = wave_live && j_hi >= j_lo ? j_hi / 2 + 1 : 0;

  const int64_t n_tiles = (a.C + kDenseCols - 1) / kDenseCols;
  const bool vec = (a.C % 4 == 0) && (a.ldx % 4 == 0) && (reinterpret_cast<uintptr_t>(a.X) % 16 == 0);
  __shared__ unsigned int s_tile;
  for (;;) {
    __syncthreads();  // previous tile's LDS reads and s_tile reads are done
    if (threadIdx.x == 0) s_tile = atomicAdd(a.tile_counter + blockIdx.y, 1u);
    __syncthreads();
    const int64_t tile = s_tile;
    if (tile >= n_tiles) break;
    const int64_t c0 = tile * kDenseCols;
    if (vec) {
      for (int t = threadIdx.x; t < a.T_in * (kDenseCols / 4); t += 256) {
        const int j = t / (kDenseCols / 4), q = t % (kDenseCols / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + 4 * q < a.C)
          v = *reinterpret_cast<const float4*>(a.X + row_pos(j, a.T_in, a.x_tl) * a.ldx + c0 + 4 * q);
        *reinterpret_cast<float4*>(&Xs[j * kDenseCols + 4 * q]) = v;
      }
    } else {
      for (int t = threadIdx.x; t < a.T_in * kDenseCols; t += 256) {
        const int j = t / kDenseCols, q = t % kDenseCols;
        Xs[t] = (c0 + q < a.C) ? a.X[row_pos(j, a.T_in, a.x_tl) * a.ldx + c0 + q] : 0.f;
      }
    }
    __syncthreads();
    if (!wave_live) continue;
#pragma unroll
    for (int nb = 0; nb < kDenseCols / 32; ++nb) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {
        if (s >= s_lo && s < s_hi) {
          const int j = 2 * s + lh;
          const float b = j < a.T_in ? Xs[j * kDenseCols + nb * 32 + li] : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], b, acc, 0, 0, 0);
        }
      }
      const int64_t c = c0 + nb * 32 + li;
      if (c < a.C) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = k0 + (i & 3) + 8 * (i >> 2) + 4 * lh;
          if (k < a.T_out) a.Y[row_pos(k, a.T_out, a.y_tl) * a.ldy + c] = acc[i];
        }
      }
    }
  }
}

// ---- dense operators on the bf16 matrix cores at fp32 accuracy (T_in <= 128) -------------------
// Minv (ehf:184, 224) and bands wider than 20 diagonals are dense T x T products against the
// [T][C] tensor: 2·T²·C flops on 8·T·C bytes — at T = 128 that is MFMA-bound on the exact-f32
// matrix instruction (kernel above).  Here both operands are split exactly into three bf16 planes
// (x = hi + mid + lo) and the six plane products that matter are formed with
// v_mfma_f32_32x32x16_bf16 (16x the f32 rate): the transform becomes a stream over X and Y.
//   tile = 64 columns x all T_in rows.  Thread (column quad t&15, row group t>>4) loads two 4x4
//   blocks (4 consecutive rows j x 4 consecutive columns: coalesced float4s along C), splits them and
//   writes, per column, the 4 j-slots it owns as one 8-byte store: the transpose to "8 consecutive j
//   per lane" (the MFMA B operand, lane = column) happens in the split pass.  LDS image
//   [plane][j/32][column quad: 272 B][column: 64 B][j%32: 2 B] — the 16-byte pad per quad makes the
//   ds_read_b128 fragment reads conflict-free.  Wave w owns output rows 32w..32w+31 of the block's
//   128-row slab; its rows of Mop live in registers as pre-split A fragments, and j-steps outside
//   the band of those rows are skipped (a triangular operator costs half the MFMAs).  Output
//   D[k][c]: lane = column, so every store instruction writes two full 128-B row segments.
//   Reduction length T_in <= 128: 48 MFMA accumulations per output, so the truncating bf16-MFMA
//   accumulator (gemm.hip, X3_FLUSH) costs < 0.1 ulp here.
typedef __bf16 mx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mx_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mx_pack(float a, float b) {  // bf16(a) | bf16(b) << 16, RNE
  mx_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mx_bf16x2));
}
__device__ __forceinline__ void mx_split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = mx_pack(a, b);
  a -= __uint_as_float(h << 16);
  b -= __uint_as_float(h & 0xffff0000u);
  m = mx_pack(a, b);
  a -= __uint_as_float(m << 16);
  b -= __uint_as_float(m & 0xffff0000u);
  l = mx_pack(a, b);
}
__device__ __forceinline__ float mx_comp(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }

constexpr int MX_COLS = 64;
constexpr int MX_QPITCH = 272;                        // bytes per column quad: 4 x 64 + 16
constexpr int MX_JB = (MX_COLS / 4) * MX_QPITCH;      // one block of 32 rows j
constexpr int MX_PLANE = 4 * MX_JB;                   // T_in <= 128

__global__ __launch_bounds__(256, 2) void mtransform_bf16x3_kernel(MtArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[3 * MX_PLANE];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int k0 = (blockIdx.y * 4 + wave) * 32;  // this wave's output rows
  const bool wave_live = k0 < a.T_out;
  const int d_lo = (a.row_off - a.col_off) - a.band_lo;
  const int d_hi = (a.row_off - a.col_off) + a.band_hi;
  // j-steps (16 rows each) that this wave's 32 output rows can touch
  int j_lo = k0 + d_lo, j_hi = k0 + 31 + d_hi;
  if (j_lo < 0) j_lo = 0;
  if (j_hi > a.T_in - 1) j_hi = a.T_in - 1;
  const int js_lo = j_lo / 16, js_hi = (wave_live && j_hi >= j_lo) ? j_hi / 16 + 1 : 0;

  // A fragments: lane (row k0+li, j = 16 js + 8 lh + e), pre-split; zero outside the operator / the band
  unsigned am[8][3][4];
  {
    const int k = k0 + li;
    const int kc = k < a.T_out ? k : a.T_out - 1;
#pragma unroll
    for (int js = 0; js < 8; ++js)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = 16 * js + 8 * lh + 2 * e + u;
          const int jc = j < a.T_in ? j : a.T_in - 1;
          const bool in = k < a.T_out && j < a.T_in && j >= k + d_lo && j <= k + d_hi;
          v[u] = mop(a, kc, jc) * (in ? 1.f : 0.f);
        }
        mx_split3(v[0], v[1], am[js][0][e], am[js][1][e], am[js][2][e]);
      }
  }

  // staging role: column quad cq, row blocks 4(jg + 16 b) .. +3, b = 0, 1
  const int cq = threadIdx.x & 15, jg = threadIdx.x >> 4;
  unsigned char* wr0 = sm + (jg >> 3) * MX_JB + cq * MX_QPITCH + (jg & 7) * 8;  // block b adds 2 j-blocks
  const unsigned char* rd = sm + (li >> 2) * MX_QPITCH + (li & 3) * 64 + lh * 16;

  // Two staging sets of 8 float4s in RESERVED registers (v192..v255; async_stage.h explains why they
  // must not be ordinary asm outputs), filled by inline-asm loads that the compiler's wait-count pass
  // does not see: the columns of tile i+2 are requested as soon as tile i has been split and are
  // waited for with a counted s_waitcnt one tile later (vmcnt(8): all but the 8 loads requested
  // since) — ~2 x 32 KB per block stay in flight across the barriers.  Same scheme as gemm_bf16x3_kernel.
  const int64_t last_quad = a.C - 4;  // C % 4 == 0 (checked by the launcher)
  // Row positions are recomputed per tile from an opaque copy of jg: hoisted out of the tile loop
  // they would pin 16 VGPRs of 64-bit offsets (plus 32 more for the stores below) beside the
  // 96-VGPR operator strip.
  auto fetch = [&](auto set, unsigned tile) __attribute__((always_inline)) {  // no conditional load: columns past C re-read the last quad, rows past T_in the last row; zeroed at the split
    constexpr int SET = decltype(set)::value;
    int64_t c = (int64_t)tile * MX_COLS + 4 * cq;
    if (c > last_quad) c = last_quad;
    const float* base = a.X + c;
    int jq = jg;
    asm volatile("" : "+v"(jq));
    auto rowp = [&](int b, int i) __attribute__((always_inline)) {
      const int j = 4 * (jq + 16 * b) + i;
      return base + row_pos(j < a.T_in ? j : a.T_in - 1, a.T_in, a.x_tl) * a.ldx;
    };
    stage8_load<SET, 0>(rowp(0, 0));
    stage8_load<SET, 1>(rowp(0, 1));
    stage8_load<SET, 2>(rowp(0, 2));
    stage8_load<SET, 3>(rowp(0, 3));
    stage8_load<SET, 4>(rowp(1, 0));
    stage8_load<SET, 5>(rowp(1, 1));
    stage8_load<SET, 6>(rowp(1, 2));
    stage8_load<SET, 7>(rowp(1, 3));
  };
  auto landed = [&](bool newer_in_flight) __attribute__((always_inline)) {  // the OLDER set's 8 loads are complete
    if (newer_in_flight)
      TMGCN_WAIT_VM(8);
    else
      TMGCN_WAIT_VM(0);
  };
  auto zrow = [&](int b, int i, float zc) __attribute__((always_inline)) { return (4 * (jg + 16 * b) + i < a.T_in) ? zc : 0.f; };  // 0 outside T_in / C
  auto split_block = [&](const stage_f32x4& s0, const stage_f32x4& s1, const stage_f32x4& s2, const stage_f32x4& s3,
                         int b) __attribute__((always_inline)) {  // the four rows arrive masked (stage8_read_mul)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h0, m0, l0, h1, m1, l1;
      mx_split3(s0[e], s1[e], h0, m0, l0);
      mx_split3(s2[e], s3[e], h1, m1, l1);
      unsigned char* w = wr0 + b * 2 * MX_JB + e * 64;
      *reinterpret_cast<uint2*>(w) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(w + MX_PLANE) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(w + 2 * MX_PLANE) = make_uint2(l0, l1);
    }
  };
  auto split_store = [&](auto set, unsigned tile) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const float zc = ((int64_t)tile * MX_COLS + 4 * cq < a.C) ? 1.f : 0.f;
    split_block(stage8_read_mul<SET, 0>(zrow(0, 0, zc)), stage8_read_mul<SET, 1>(zrow(0, 1, zc)),
                stage8_read_mul<SET, 2>(zrow(0, 2, zc)), stage8_read_mul<SET, 3>(zrow(0, 3, zc)), 0);
    split_block(stage8_read_mul<SET, 4>(zrow(1, 0, zc)), stage8_read_mul<SET, 5>(zrow(1, 1, zc)),
                stage8_read_mul<SET, 6>(zrow(1, 2, zc)), stage8_read_mul<SET, 7>(zrow(1, 3, zc)), 1);
  };
  const std::integral_constant<int, 0> SA;
  const std::integral_constant<int, 1> SB;

  f32x16 acc[2];
  struct Frag3 {
    uint4 h, m, l;
  };
  auto read_group = [&](int js, int cb) __attribute__((always_inline)) {  // X-plane fragments of j-step js, column block cb
    const unsigned char* p = rd + (js >> 1) * MX_JB + cb * 8 * MX_QPITCH + (js & 1) * 32;
    Frag3 f;
    f.h = *reinterpret_cast<const uint4*>(p);
    f.m = *reinterpret_cast<const uint4*>(p + MX_PLANE);
    f.l = *reinterpret_cast<const uint4*>(p + 2 * MX_PLANE);
    return f;
  };
  auto six = [&](int js, int cb, const Frag3& f) __attribute__((always_inline)) {  // the six plane products of one (j-step, column block)
    const mx_bf16x8 ah = __builtin_bit_cast(mx_bf16x8, make_uint4(am[js][0][0], am[js][0][1], am[js][0][2], am[js][0][3]));
    const mx_bf16x8 amid = __builtin_bit_cast(mx_bf16x8, make_uint4(am[js][1][0], am[js][1][1], am[js][1][2], am[js][1][3]));
    const mx_bf16x8 al = __builtin_bit_cast(mx_bf16x8, make_uint4(am[js][2][0], am[js][2][1], am[js][2][2], am[js][2][3]));
    const mx_bf16x8 bh = __builtin_bit_cast(mx_bf16x8, f.h), bm = __builtin_bit_cast(mx_bf16x8, f.m),
                    bl = __builtin_bit_cast(mx_bf16x8, f.l);
    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[cb], 0, 0, 0);  // small terms first
    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[cb], 0, 0, 0);
    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid, bm, acc[cb], 0, 0, 0);
    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(amid, bh, acc[cb], 0, 0, 0);
    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[cb], 0, 0, 0);
    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[cb], 0, 0, 0);
  };
  // j-steps [LO, HI) known at compile time: groups (j-step, column block) software-pipelined by hand —
  // the three fragment reads of group g+1 are issued in front of the six MFMAs of group g
  // (sched_barrier pins the order), so a read has 192 cycles of matrix-pipe time to land instead of
  // being waited for right in front of its MFMA.
  auto pipelined = [&](auto lo_tag, auto hi_tag) __attribute__((always_inline)) {
    constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value, NG = 2 * (HI - LO);
    Frag3 cur = read_group(LO, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      Frag3 nxt = cur;
      if (g + 1 < NG) nxt = read_group(LO + ((g + 1) >> 1), (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      six(LO + (g >> 1), g & 1, cur);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  };
  auto multiply = [&]() __attribute__((always_inline)) {
    if (!wave_live) return;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
    using std::integral_constant;
    // the ranges a dense / lower- / upper-triangular 128-row operator gives the four waves of a block
    const int range = js_lo * 16 + js_hi;
    switch (range) {
      case 0 * 16 + 8: pipelined(integral_constant<int, 0>{}, integral_constant<int, 8>{}); return;
      case 0 * 16 + 6: pipelined(integral_constant<int, 0>{}, integral_constant<int, 6>{}); return;
      case 0 * 16 + 4: pipelined(integral_constant<int, 0>{}, integral_constant<int, 4>{}); return;
      case 0 * 16 + 2: pipelined(integral_constant<int, 0>{}, integral_constant<int, 2>{}); return;
      case 2 * 16 + 8: pipelined(integral_constant<int, 2>{}, integral_constant<int, 8>{}); return;
      case 4 * 16 + 8: pipelined(integral_constant<int, 4>{}, integral_constant<int, 8>{}); return;
      case 6 * 16 + 8: pipelined(integral_constant<int, 6>{}, integral_constant<int, 8>{}); return;
      default: break;
    }
#pragma unroll
    for (int js = 0; js < 8; ++js) {  // any other band / size: not pipelined
      if (js >= js_lo && js < js_hi) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) six(js, cb, read_group(js, cb));
      }
    }
  };
  // D[k][c]: column = lane & 31, row = (i&3) + 8*(i>>2) + 4*(lane>>5).  Each 4x4 block (4 rows in
  // registers x 4 columns on a lane quad) is transposed in registers (quad_transpose4, common.h):
  // lane j of a quad then owns row j and four consecutive columns — one 16-byte store per lane
  // instead of four dword stores (the dword epilogue is bound by store issue, not bandwidth).
  auto store_tile = [&](unsigned tile) __attribute__((always_inline)) {
    if (!wave_live) return;
    const int64_t c0 = (int64_t)tile * MX_COLS;
    int lq = li, kb = k0 + 4 * lh;
    asm volatile("" : "+v"(lq), "+v"(kb));  // see fetch(): keep the row offsets out of the loop-invariant set
    const int j = lq & 3, q = lq >> 2;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int64_t c = c0 + cb * 32 + 4 * q;  // C % 4 == 0: a column quad is inside or outside as a whole
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4] = {acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3]};
        quad_transpose4(v, j);
        const int k = kb + 8 * g + j;
        if (k < a.T_out && c < a.C)
          *reinterpret_cast<float4*>(a.Y + row_pos(k, a.T_out, a.y_tl) * a.ldy + c) = make_float4(v[0], v[1], v[2], v[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // static persistent schedule: block b takes column tiles b, b + G, b + 2G, ... (uniform tiles);
  // tiles alternate between the two staging sets, so the loop body is written for two
  const unsigned n_tiles = (unsigned)((a.C + MX_COLS - 1) / MX_COLS), stride = gridDim.x;
  unsigned t0 = blockIdx.x, t1 = blockIdx.x + stride;
  if (t0 < n_tiles) fetch(SA, t0);
  if (t1 < n_tiles) fetch(SB, t1);
  landed(t1 < n_tiles);  // set A
  while (t0 < n_tiles) {
    // ---- tile t0 (set A): split, request t0 + 2G into set A, multiply, wait for set B, store
    __syncthreads();  // the previous tile's fragment reads are done
    split_store(SA, t0);
    const unsigned t2 = t0 + 2 * stride;
    const bool f2 = t2 < n_tiles;
    if (f2) fetch(SA, t2);
    __syncthreads();
    multiply();
    landed(f2);  // set B
    store_tile(t0);
    if (t1 >= n_tiles) break;
    // ---- tile t1 (set B)
    __syncthreads();
    split_store(SB, t1);
    const unsigned t3 = t1 + 2 * stride;
    const bool f3 = t3 < n_tiles;
    if (f3) fetch(SB, t3);
    __syncthreads();
    multiply();
    landed(f3);  // set A
    store_tile(t1);
    t0 = t2;
    t1 = t3;
  }
}

template <int WIDTH, int VEC>
static void launch_band(const MtArgs& a, dim3 grid, hipStream_t st) {
  constexpr int PF = 4;  // rows in flight per lane
  const size_t smem = (size_t)a.rows_per_chunk * WIDTH * sizeof(float);
  if (a.x_tl || a.y_tl)
    hipLaunchKernelGGL((mtransform_band_kernel<WIDTH, PF, VEC, true>), grid, dim3(256), smem, st, a);
  else
    hipLaunchKernelGGL((mtransform_band_kernel<WIDTH, PF, VEC, false>), grid, dim3(256), smem, st, a);
}

template <int VEC>
static int dispatch(MtArgs a, hipStream_t st) {
  const int64_t cvec = (a.C + VEC - 1) / VEC;
  const int64_t width = (int64_t)a.band_lo + a.band_hi + 1;
  if (width <= 20) {  // wider windows do not fit the register file unrolled: dense kernel
    const unsigned col_blocks = (unsigned)((cvec + 255) / 256);
    // chunk the output rows only when the column grid alone cannot fill 256 CUs
    int chunks = 1;
    if (col_blocks < 2048) {
      chunks = (int)((2048 + col_blocks - 1) / col_blocks);
      const int max_chunks = (a.T_out + 7) / 8;
      if (chunks > max_chunks) chunks = max_chunks;
      if (chunks < 1) chunks = 1;
    }
    const int min_chunks = (a.T_out + kBandMaxChunkRows - 1) / kBandMaxChunkRows;
    if (chunks < min_chunks) chunks = min_chunks;
    a.rows_per_chunk = (a.T_out + chunks - 1) / chunks;
    chunks = (a.T_out + a.rows_per_chunk - 1) / a.rows_per_chunk;
    dim3 grid(col_blocks, chunks);
    if (width <= 1) launch_band<1, VEC>(a, grid, st);
    else if (width <= 2) launch_band<2, VEC>(a, grid, st);
    else if (width <= 4) launch_band<4, VEC>(a, grid, st);
    else if (width <= 8) launch_band<8, VEC>(a, grid, st);
    else if (width <= 12) launch_band<12, VEC>(a, grid, st);
    else if (width <= 16) launch_band<16, VEC>(a, grid, st);
    else launch_band<20, VEC>(a, grid, st);
    return check_launch("mtransform_band");
  }
  if (a.T_in <= 128 && VEC == 4) {  // bf16 matrix cores, exact 3-way split (fp32-accurate): a stream over X and Y
    const unsigned gy = (unsigned)((a.T_out + 127) / 128);
    const int64_t n_tiles = (a.C + MX_COLS - 1) / MX_COLS;
    if (gy > 64 || n_tiles >= (int64_t)0x7fffffff) {
      set_error("mtransform: shape too large for the dense tile scheduler");
      return TMGCN_ERR_INVALID;
    }
    int64_t gx = persistent_grid(mtransform_bf16x3_kernel, 256);  // static persistent schedule: no tile counter
    if (gx > n_tiles) gx = n_tiles;
    hipLaunchKernelGGL(mtransform_bf16x3_kernel, dim3((unsigned)gx, gy), dim3(256), 0, st, a);
    return check_launch("mtransform_bf16x3");
  }
  if (a.T_in <= 256) {  // exact-f32 matrix-core path (129 <= T_in <= 256, or unaligned / C % 4 != 0)
    const size_t smem = (size_t)a.T_in * kDenseCols * sizeof(float);
    const unsigned gy = (unsigned)((a.T_out + 127) / 128);
    const int64_t n_tiles = (a.C + kDenseCols - 1) / kDenseCols;
    if (gy > 64 || n_tiles >= (int64_t)0x7fffffff) {
      set_error("mtransform: shape too large for the dense tile scheduler");
      return TMGCN_ERR_INVALID;
    }
    a.tile_counter = acquire_tile_counters(st, (int)gy);
    if (!a.tile_counter) {
      set_error("mtransform: no tile counters: %s", pool_error());
      return TMGCN_ERR_LAUNCH;
    }
    if (a.T_in <= 128) {
      int64_t gx = persistent_grid(mtransform_dense_mfma_kernel<64>, 256, smem);
      if (gx > n_tiles) gx = n_tiles;
      hipLaunchKernelGGL((mtransform_dense_mfma_kernel<64>), dim3((unsigned)gx, gy), dim3(256), smem, st, a);
    } else {
      int64_t gx = persistent_grid(mtransform_dense_mfma_kernel<128>, 256, smem);
      if (gx > n_tiles) gx = n_tiles;
      hipLaunchKernelGGL((mtransform_dense_mfma_kernel<128>), dim3((unsigned)gx, gy), dim3(256), smem, st, a);
    }
    return check_launch("mtransform_dense_mfma");
  }
  constexpr int RT = 16;
  dim3 grid((unsigned)((cvec + kWave - 1) / kWave), (unsigned)((a.T_out + 4 * RT - 1) / (4 * RT)));
  hipLaunchKernelGGL((mtransform_dense_kernel<RT, VEC>), grid, dim3(256), 0, st, a);
  return check_launch("mtransform_dense");
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_mtransform_ld_f32(const float* M, int32_t Tm, int32_t ldm, int32_t transpose,
                                        int32_t row_off, int32_t col_off, int32_t T_out,
                                        int32_t T_in, int32_t band_lo, int32_t band_hi,
                                        const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t C,
                                        int32_t x_group_rows, int32_t y_group_rows, void* stream) {
  TMGCN_REQUIRE(Tm > 0 && ldm >= Tm, "mtransform: bad operator shape Tm=%d ldm=%d", Tm, ldm);
  TMGCN_REQUIRE(T_out >= 0 && T_in >= 0 && C >= 0, "mtransform: negative extent");
  TMGCN_REQUIRE(ldx >= C && ldy >= C, "mtransform: row strides ldx=%lld ldy=%lld are smaller than C=%lld", (long long)ldx,
                (long long)ldy, (long long)C);
  TMGCN_REQUIRE(row_off >= 0 && col_off >= 0 && row_off + T_out <= Tm && col_off + T_in <= Tm,
                "mtransform: window [%d+%d) x [%d+%d) exceeds the %dx%d operator", row_off,
                T_out, col_off, T_in, Tm, Tm);
  TMGCN_REQUIRE(band_lo >= 0 && band_hi >= 0, "mtransform: negative band");
  if (T_out == 0 || C == 0) return TMGCN_OK;
  TMGCN_REQUIRE(M && X && Y, "mtransform: null pointer");
  TMGCN_REQUIRE(X != Y, "mtransform: in-place transform is not supported");
  TMGCN_REQUIRE(x_group_rows >= 0 && (x_group_rows == 0 || T_in % x_group_rows == 0),
                "mtransform: x_group_rows=%d does not divide T_in=%d", x_group_rows, T_in);
  TMGCN_REQUIRE(y_group_rows >= 0 && (y_group_rows == 0 || T_out % y_group_rows == 0),
                "mtransform: y_group_rows=%d does not divide T_out=%d", y_group_rows, T_out);
  if (band_lo > Tm) band_lo = Tm;
  if (band_hi > Tm) band_hi = Tm;
  MtArgs a{M, ldm, transpose, row_off, col_off, T_out, T_in, band_lo, band_hi, X, Y, C, T_out, x_group_rows, y_group_rows, nullptr, ldx, ldy};
  const bool vec_ok = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (reinterpret_cast<uintptr_t>(X) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(Y) % 16 == 0);
  return vec_ok ? dispatch<4>(a, (hipStream_t)stream) : dispatch<1>(a, (hipStream_t)stream);
}

extern "C" int tmgcn_mtransform_f32(const float* M, int32_t Tm, int32_t ldm, int32_t transpose,
                                     int32_t row_off, int32_t col_off, int32_t T_out,
                                     int32_t T_in, int32_t band_lo, int32_t band_hi,
                                     const float* X, float* Y, int64_t C, int32_t x_group_rows,
                                     int32_t y_group_rows, void* stream) {
  return tmgcn_mtransform_ld_f32(M, Tm, ldm, transpose, row_off, col_off, T_out, T_in, band_lo, band_hi, X, C, Y, C, C,
                                 x_group_rows, y_group_rows, stream);
}

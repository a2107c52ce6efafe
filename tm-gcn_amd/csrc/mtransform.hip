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
//   mtransform_dense    any operator (e.g. Minv): register-blocked 16 rows x float4 per lane.
#include "common.h"

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
};
template <>
struct Cols<1> {
  using T = float;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, const T& v) { *p = v; }
  static __device__ __forceinline__ void fma(T& acc, float m, const T& x) { acc = fmaf(m, x, acc); }
};

// Sliding-window band kernel.  Window slot of input row j is (j + JB) mod W with JB chosen
// so that slots are compile-time constants inside the unrolled body.
template <int W, int VEC>
__global__ __launch_bounds__(256) void mtransform_band_kernel(MtArgs a) {
  using CT = Cols<VEC>;
  using V = typename CT::T;
  const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (c >= a.C) return;
  const int k_begin = blockIdx.y * a.rows_per_chunk;
  int k_end = k_begin + a.rows_per_chunk;
  if (k_end > a.T_out) k_end = a.T_out;
  if (k_begin >= k_end) return;

  // output row k reads input rows j in [k + d_lo, k + d_hi]
  const int d_lo = (a.row_off - a.col_off) - a.band_lo;
  const int d_hi = (a.row_off - a.col_off) + a.band_hi;  // d_hi - d_lo + 1 <= W

  V win[W];
#pragma unroll
  for (int i = 0; i < W; ++i) win[i] = CT::zero();

  // q walks the input rows, starting at the oldest row the first output row needs.
  // q is kept congruent to the unrolled index i modulo W:  q = qb + i.
  const int q_first = k_begin + d_lo;
  // floor to a multiple of W (q_first may be negative)
  int qb = q_first >= 0 ? (q_first / W) * W : -(((-q_first) + W - 1) / W) * W;
  const int q_last = k_end - 1 + d_hi;

  for (; qb <= q_last; qb += W) {
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const int q = qb + i;  // slot i  <->  input row q  (q mod W == i since qb % W == 0)
      if (q >= q_first && q <= q_last) {
        win[i] = (q >= 0 && q < a.T_in) ? CT::load(a.X + row_pos(q, a.T_in, a.x_tl) * a.C + c) : CT::zero();
        const int k = q - d_hi;  // output row completed by this input row
        if (k >= k_begin) {
          V acc = CT::zero();
#pragma unroll
          for (int d = 0; d < W; ++d) {
            // input row j = q - d lives in slot (i - d) mod W
            const int j = q - d;
            if (j >= k + d_lo && j >= 0 && j < a.T_in) {
              CT::fma(acc, mop(a, k, j), win[(i - d + W) % W]);
            }
          }
          CT::store(a.Y + row_pos(k, a.T_out, a.y_tl) * a.C + c, acc);
        }
      }
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
    const V x = CT::load(a.X + row_pos(j, a.T_in, a.x_tl) * a.C + c);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int k = k0 + i;
      if (k < a.T_out && j >= k + d_lo && j <= k + d_hi) CT::fma(acc[i], mop(a, k, j), x);
    }
  }
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int k = k0 + i;
    if (k < a.T_out) CT::store(a.Y + row_pos(k, a.T_out, a.y_tl) * a.C + c, acc[i]);
  }
}

template <int W, int VEC>
static void launch_band(const MtArgs& a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL((mtransform_band_kernel<W, VEC>), grid, dim3(256), 0, st, a);
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
  constexpr int RT = 16;
  dim3 grid((unsigned)((cvec + kWave - 1) / kWave), (unsigned)((a.T_out + 4 * RT - 1) / (4 * RT)));
  hipLaunchKernelGGL((mtransform_dense_kernel<RT, VEC>), grid, dim3(256), 0, st, a);
  return check_launch("mtransform_dense");
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_mtransform_f32(const float* M, int32_t Tm, int32_t ldm, int32_t transpose,
                                     int32_t row_off, int32_t col_off, int32_t T_out,
                                     int32_t T_in, int32_t band_lo, int32_t band_hi,
                                     const float* X, float* Y, int64_t C, int32_t x_group_rows,
                                     int32_t y_group_rows, void* stream) {
  TMGCN_REQUIRE(Tm > 0 && ldm >= Tm, "mtransform: bad operator shape Tm=%d ldm=%d", Tm, ldm);
  TMGCN_REQUIRE(T_out >= 0 && T_in >= 0 && C >= 0, "mtransform: negative extent");
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
  MtArgs a{M, ldm, transpose, row_off, col_off, T_out, T_in, band_lo, band_hi, X, Y, C, T_out, x_group_rows, y_group_rows};
  const bool vec_ok = (C % 4 == 0) && (reinterpret_cast<uintptr_t>(X) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(Y) % 16 == 0);
  return vec_ok ? dispatch<4>(a, (hipStream_t)stream) : dispatch<1>(a, (hipStream_t)stream);
}

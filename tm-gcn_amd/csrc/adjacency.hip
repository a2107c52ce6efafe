// Adjacency pipeline on the device: raw dynamic-graph edges -> normalised, M-transformed
// batched CSR   (gfx950 / CDNA4)
//
// Replaces the offline preprocessing the reference does with per-slice / per-nnz Python loops
// (minutes): read_data.py:88-111 func_make_symmetric, :116-125 func_edge_life, :130-169
// func_laplacian_transformation, :204-223 func_MProduct (MATLAB: read_data.m:172-209), and the
// COO ingest of ehf:560-574.
//
// Every step is the same primitive on a batched COO whose entries carry ONE 64-bit key
//      key = (slice * N + row) * N + col
//   expand   each entry fans out to a few entries (transpose copy / later slices of the edge-life
//            window / the slices k with M[k, j] != 0), one thread per output entry
//   sort     rocPRIM radix sort of (key, value) pairs
//   reduce   entries with equal keys are summed in sorted order (fixed order: reproducible)
// followed by elementwise normalisation with the row sums.  Keys double as CSR: rowptr is a
// binary search of r*N in the sorted keys, col = key mod N.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/functional.hpp>

#include "common.h"

namespace tmgcn {

// ---- expand kernels ---------------------------------------------------------------------
// symmetrise: out[2p] = (t,i,j, v/2), out[2p+1] = (t,j,i, v/2)         (read_data.py:97-99)
__global__ void adj_symmetrise_kernel(const uint64_t* __restrict__ key, const float* __restrict__ val,
                                      int64_t n, int64_t N, uint64_t* __restrict__ okey,
                                      float* __restrict__ oval) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint64_t k = key[p];
  const uint64_t col = k % N, tr = k / N;   // tr = slice*N + row
  const uint64_t row = tr % N, t = tr / N;
  const float v = 0.5f * val[p];
  okey[2 * p] = k;
  oval[2 * p] = v;
  okey[2 * p + 1] = (t * N + col) * N + row;
  oval[2 * p + 1] = v;
}

// edge life: entry of slice t also lives in slices t+1 .. t+L-1 (< T)   (read_data.py:116-125)
// out has n*L slots; slots past the last slice get the sentinel key ~0 and value 0
__global__ void adj_edge_life_kernel(const uint64_t* __restrict__ key, const float* __restrict__ val,
                                     int64_t n, int64_t N, int32_t T, int32_t L,
                                     uint64_t* __restrict__ okey, float* __restrict__ oval) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n * L) return;
  const int64_t p = q / L;
  const int s = (int)(q % L);
  const uint64_t k = key[p];
  const uint64_t nn = (uint64_t)N * N;
  const uint64_t t = k / nn;
  if (t + s < (uint64_t)T) {
    okey[q] = k + (uint64_t)s * nn;
    oval[q] = val[p];
  } else {
    okey[q] = ~0ull;
    oval[q] = 0.f;
  }
}

// identity entries (t, i, i, 1) for every slice and node, appended at okey[0 .. T*N)
__global__ void adj_identity_kernel(int64_t TN, int64_t N, uint64_t* __restrict__ okey,
                                    float* __restrict__ oval) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= TN) return;
  okey[r] = (uint64_t)r * N + (uint64_t)(r % N);
  oval[r] = 1.f;
}

// row sums of a sorted COO -> d[r] = 1/sqrt(sum)          (read_data.py:146-147)
__global__ void adj_rowsum_kernel(const uint64_t* __restrict__ key, const float* __restrict__ val,
                                  const int64_t* __restrict__ rowptr, int64_t TN,
                                  float* __restrict__ dinv) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= TN) return;
  double s = 0.0;
  for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) s += (double)val[p];
  dinv[r] = s > 0.0 ? (float)(1.0 / sqrt(s)) : 0.f;
}

// v *= d[slice*N+row] * d[slice*N+col]                     (read_data.py:157-159)
__global__ void adj_scale_kernel(const uint64_t* __restrict__ key, float* __restrict__ val, int64_t n,
                                 int64_t N, const float* __restrict__ dinv) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint64_t k = key[p];
  const uint64_t tr = k / N, col = k % N;
  const uint64_t t = tr / N;
  val[p] = (float)((double)val[p] * (double)dinv[tr] * (double)dinv[t * N + col]);
}

// M-product: entry (j, r, c, v) fans out to (k, r, c, M[k][j]*v) for the W slices k = j + koff[s],
// s = 0..W-1, where the band of M (lower lo, upper hi) bounds the non-zeros of column j
// (read_data.py:204-223).  Slots with k outside [0,T) or M[k][j] == 0 get the sentinel.
__global__ void adj_mproduct_kernel(const uint64_t* __restrict__ key, const float* __restrict__ val,
                                    int64_t n, int64_t N, int32_t T, const float* __restrict__ M,
                                    int32_t ldm, int32_t lo, int32_t hi,
                                    uint64_t* __restrict__ okey, float* __restrict__ oval) {
  const int W = lo + hi + 1;
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n * W) return;
  const int64_t p = q / W;
  const int s = (int)(q % W);
  const uint64_t k0 = key[p];
  const uint64_t nn = (uint64_t)N * N;
  const int64_t j = (int64_t)(k0 / nn);
  const int64_t k = j - hi + s;  // rows k of M with M[k][j] possibly non-zero: j-hi .. j+lo
  float m = 0.f;
  if (k >= 0 && k < T) m = M[k * ldm + j];
  if (m != 0.f) {
    okey[q] = k0 + (uint64_t)(k - j) * nn;
    oval[q] = m * val[p];
  } else {
    okey[q] = ~0ull;
    oval[q] = 0.f;
  }
}

// ---- CSR views of sorted keys ----------------------------------------------------------------
// rowptr from sorted keys in O(nnz): entry p owns the rows (row(p-1), row(p)] — every row that
// starts at p — and the last entry also closes the tail; with n == 0 thread 0 fills everything.
__global__ void adj_rowptr_kernel(const uint64_t* __restrict__ key, int64_t n, int64_t N, int64_t TN,
                                  int64_t* __restrict__ rowptr) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0) {
    for (int64_t r = p; r <= TN; r += (int64_t)gridDim.x * blockDim.x) rowptr[r] = 0;
    return;
  }
  if (p >= n) return;
  const int64_t row = (int64_t)(key[p] / (uint64_t)N);
  const int64_t prev = p ? (int64_t)(key[p - 1] / (uint64_t)N) : -1;
  for (int64_t r = prev + 1; r <= row; ++r) rowptr[r] = p;   // rows (prev, row] begin at p
  if (p == n - 1)
    for (int64_t r = row + 1; r <= TN; ++r) rowptr[r] = n;   // empty tail rows and the end marker
}

__global__ void adj_cols_kernel(const uint64_t* __restrict__ key, int64_t n, int64_t N,
                                int32_t* __restrict__ col) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  col[p] = (int32_t)(key[p] % N);
}

__global__ void adj_make_keys_kernel(const int64_t* __restrict__ t, const int64_t* __restrict__ i,
                                     const int64_t* __restrict__ j, int64_t n, int64_t N,
                                     uint64_t* __restrict__ key) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  key[p] = ((uint64_t)t[p] * N + (uint64_t)i[p]) * N + (uint64_t)j[p];
}

// transposed key of a CSR entry: (slice, col, row)
__global__ void adj_transpose_keys_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                          int64_t TN, int64_t N, uint64_t* __restrict__ okey) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= TN) return;
  const uint64_t t = r / N, i = r % N;
  for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) okey[p] = (t * N + (uint64_t)col[p]) * N + i;
}

// ---- M-product as a segmented merge (no expansion, no sort) -------------------------------------
// Output row (k, r) = Σ_j M[k][j] · A_j[r, :] over the slices j the band of M reaches from k.  The
// input rows are column-sorted CSR rows, so the output row is their W-way merge: a group of L lanes
// (32: two rows per wave, or 64) owns one output row, lane l walks the row of slice j = k - lo + l;
// every step the group takes the smallest head column (wave-shuffle min), the lanes standing on it
// contribute m·val (summed in fp64 in a fixed lane order, rounded once) and advance.  Two passes over
// the same walk: COUNT (row lengths -> the caller's prefix sum gives rowptr) and FILL.  Memory: the
// output itself; the expand + sort form (kept as the fallback for bands wider than 64) needs
// W x nnz x 12 B of keys and values plus the sort's double buffer.
template <int L, bool FILL>
__global__ __launch_bounds__(256) void adj_mproduct_merge_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    int64_t TN, int32_t N, int32_t T, const float* __restrict__ M, int32_t ldm, int32_t lo, int32_t hi,
    int64_t* __restrict__ out_count, const int64_t* __restrict__ out_rowptr, int32_t* __restrict__ out_col,
    float* __restrict__ out_val) {
  constexpr int kNone = 0x7fffffff;
  const int lane = threadIdx.x & (L - 1);
  const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / L;   // output row k*N + r
  if (row >= TN) return;                                                       // whole groups leave together
  const int64_t k = row / N, r = row - k * N;
  const int64_t j = k - lo + lane;
  float m = 0.f;
  if (lane <= lo + hi && j >= 0 && j < T) m = M[k * ldm + j];
  int64_t p = 0, e = 0;
  if (m != 0.f) {
    p = rowptr[j * N + r];
    e = rowptr[j * N + r + 1];
  }
  int c = p < e ? col[p] : kNone;
  int64_t o = FILL ? out_rowptr[row] : 0;
  int64_t n_out = 0;
  for (;;) {
    int cmin = c;
#pragma unroll
    for (int s = L >> 1; s > 0; s >>= 1) {
      const int other = __shfl_xor(cmin, s, L);
      cmin = other < cmin ? other : cmin;
    }
    if (cmin == kNone) break;
    double acc = 0.0;
    if (c == cmin) {                                 // (duplicate columns inside an input row are summed too)
      do {
        if (FILL) acc += (double)m * (double)val[p];
        ++p;
        c = p < e ? col[p] : kNone;
      } while (c == cmin);
    }
    if (FILL) {
#pragma unroll
      for (int s = L >> 1; s > 0; s >>= 1) acc += __shfl_xor(acc, s, L);   // fixed butterfly: reproducible
      if (lane == 0) {
        out_col[o] = cmin;
        out_val[o] = (float)acc;
      }
      ++o;
    }
    ++n_out;
  }
  if (!FILL && lane == 0) out_count[row + 1] = n_out;
  if (!FILL && row == 0 && lane == 0) out_count[0] = 0;
}

static inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace tmgcn

using namespace tmgcn;

// ---- sort + reduce (rocPRIM) ------------------------------------------------------------------
// Sorts (key, val) pairs by key and sums runs of equal keys.  Entries with the sentinel key ~0
// are dropped.  Outputs: keys_out/vals_out (capacity n), *n_out_dev (device int64).
// Workspace layout: [keys_sorted n*8][vals_sorted n*4][count 8][rocprim temp].
extern "C" int64_t tmgcn_coo_sort_reduce_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  size_t t1 = 0, t2 = 0;
  uint64_t* k = nullptr;
  float* v = nullptr;
  int64_t* c = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, t1, k, k, v, v, (size_t)n, 0, 64, (hipStream_t)0);
  (void)rocprim::reduce_by_key(nullptr, t2, k, v, (size_t)n, k, v, c, rocprim::plus<float>(),
                               rocprim::equal_to<uint64_t>(), (hipStream_t)0);
  const size_t tmp = t1 > t2 ? t1 : t2;
  return (int64_t)(align256((size_t)n * 8) + align256((size_t)n * 4) + 256 + align256(tmp));
}

extern "C" int tmgcn_coo_sort_reduce(const uint64_t* keys_in, const float* vals_in, int64_t n,
                                      int32_t key_bits, uint64_t* keys_out, float* vals_out,
                                      int64_t* n_out_dev, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
  TMGCN_REQUIRE(n >= 0, "coo_sort_reduce: negative n");
  hipStream_t st = (hipStream_t)stream;
  TMGCN_REQUIRE(n_out_dev, "coo_sort_reduce: null n_out");
  if (n == 0) {
    (void)hipMemsetAsync(n_out_dev, 0, sizeof(int64_t), st);
    return check_launch("coo_sort_reduce memset");
  }
  TMGCN_REQUIRE(keys_in && vals_in && keys_out && vals_out && workspace, "coo_sort_reduce: null pointer");
  const int64_t need = tmgcn_coo_sort_reduce_workspace_bytes(n);
  if (workspace_bytes < need) {
    set_error("coo_sort_reduce: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
    return TMGCN_ERR_WORKSPACE;
  }
  char* w = (char*)workspace;
  uint64_t* ks = (uint64_t*)w;
  w += align256((size_t)n * 8);
  float* vs = (float*)w;
  w += align256((size_t)n * 4);
  w += 256;
  size_t tmp = (size_t)(workspace_bytes - (w - (char*)workspace));
  // the sentinel ~0 has every bit set, so all 64 bits are sorted when sentinels may be present
  const unsigned end_bit = (key_bits > 0 && key_bits < 64) ? (unsigned)key_bits : 64u;
  hipError_t e = rocprim::radix_sort_pairs(w, tmp, keys_in, ks, vals_in, vs, (size_t)n, 0, end_bit, st);
  if (e != hipSuccess) {
    set_error("coo_sort_reduce: radix sort: %s", hipGetErrorString(e));
    return TMGCN_ERR_LAUNCH;
  }
  tmp = (size_t)(workspace_bytes - (w - (char*)workspace));
  e = rocprim::reduce_by_key(w, tmp, ks, vs, (size_t)n, keys_out, vals_out, n_out_dev,
                             rocprim::plus<float>(), rocprim::equal_to<uint64_t>(), st);
  if (e != hipSuccess) {
    set_error("coo_sort_reduce: reduce_by_key: %s", hipGetErrorString(e));
    return TMGCN_ERR_LAUNCH;
  }
  return TMGCN_OK;
}

// ---- the expand / elementwise steps --------------------------------------------------------------
extern "C" int tmgcn_adj_make_keys(const int64_t* t, const int64_t* i, const int64_t* j, int64_t n,
                                    int64_t N, uint64_t* key, void* stream) {
  TMGCN_REQUIRE(n >= 0 && N > 0, "adj_make_keys: bad size");
  if (n == 0) return TMGCN_OK;
  hipLaunchKernelGGL(adj_make_keys_kernel, dim3(blocks(n)), dim3(256), 0, (hipStream_t)stream, t, i, j, n, N, key);
  return check_launch("adj_make_keys");
}

extern "C" int tmgcn_adj_symmetrise(const uint64_t* key, const float* val, int64_t n, int64_t N,
                                     uint64_t* okey, float* oval, void* stream) {
  TMGCN_REQUIRE(n >= 0 && N > 0, "adj_symmetrise: bad size");
  if (n == 0) return TMGCN_OK;
  hipLaunchKernelGGL(adj_symmetrise_kernel, dim3(blocks(n)), dim3(256), 0, (hipStream_t)stream, key, val, n, N,
                     okey, oval);
  return check_launch("adj_symmetrise");
}

extern "C" int tmgcn_adj_edge_life(const uint64_t* key, const float* val, int64_t n, int64_t N, int32_t T,
                                    int32_t window, uint64_t* okey, float* oval, void* stream) {
  TMGCN_REQUIRE(n >= 0 && N > 0 && T > 0 && window >= 1, "adj_edge_life: bad size");
  if (n == 0) return TMGCN_OK;
  hipLaunchKernelGGL(adj_edge_life_kernel, dim3(blocks(n * window)), dim3(256), 0, (hipStream_t)stream, key, val,
                     n, N, T, window, okey, oval);
  return check_launch("adj_edge_life");
}

extern "C" int tmgcn_adj_identity(int64_t TN, int64_t N, uint64_t* okey, float* oval, void* stream) {
  TMGCN_REQUIRE(TN >= 0 && N > 0, "adj_identity: bad size");
  if (TN == 0) return TMGCN_OK;
  hipLaunchKernelGGL(adj_identity_kernel, dim3(blocks(TN)), dim3(256), 0, (hipStream_t)stream, TN, N, okey, oval);
  return check_launch("adj_identity");
}

// C = D^-1/2 (B) D^-1/2 in place on a sorted, reduced COO; rowptr (TN+1) and dinv (TN) are outputs too
extern "C" int tmgcn_adj_normalise(const uint64_t* key, float* val, int64_t n, int64_t N, int64_t TN,
                                    int64_t* rowptr, float* dinv, void* stream) {
  TMGCN_REQUIRE(n >= 0 && N > 0 && TN >= 0, "adj_normalise: bad size");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adj_rowptr_kernel, dim3(blocks(n ? n : TN + 1)), dim3(256), 0, st, key, n, N, TN, rowptr);
  if (TN) hipLaunchKernelGGL(adj_rowsum_kernel, dim3(blocks(TN)), dim3(256), 0, st, key, val, rowptr, TN, dinv);
  if (n) hipLaunchKernelGGL(adj_scale_kernel, dim3(blocks(n)), dim3(256), 0, st, key, val, n, N, dinv);
  return check_launch("adj_normalise");
}

extern "C" int tmgcn_adj_mproduct_expand(const uint64_t* key, const float* val, int64_t n, int64_t N,
                                          int32_t T, const float* M, int32_t ldm, int32_t band_lo,
                                          int32_t band_hi, uint64_t* okey, float* oval, void* stream) {
  TMGCN_REQUIRE(n >= 0 && N > 0 && T > 0 && band_lo >= 0 && band_hi >= 0, "adj_mproduct: bad size");
  if (n == 0) return TMGCN_OK;
  TMGCN_REQUIRE(M && ldm >= T, "adj_mproduct: bad M");
  if (band_lo > T - 1) band_lo = T - 1;
  if (band_hi > T - 1) band_hi = T - 1;
  const int W = band_lo + band_hi + 1;
  hipLaunchKernelGGL(adj_mproduct_kernel, dim3(blocks(n * W)), dim3(256), 0, (hipStream_t)stream, key, val, n, N,
                     T, M, ldm, band_lo, band_hi, okey, oval);
  return check_launch("adj_mproduct");
}

// M-product of a batched CSR by segmented merge: pass 1 (row lengths), pass 2 (columns and values)
template <bool FILL>
static int mproduct_merge_launch(const int64_t* rowptr, const int32_t* col, const float* val, int64_t TN, int32_t N,
                                 int32_t T, const float* M, int32_t ldm, int32_t lo, int32_t hi, int64_t* out_count,
                                 const int64_t* out_rowptr, int32_t* out_col, float* out_val, hipStream_t st) {
  TMGCN_REQUIRE(TN >= 0 && N > 0 && T > 0 && lo >= 0 && hi >= 0, "adj_mproduct_merge: bad size");
  TMGCN_REQUIRE(TN == (int64_t)T * N, "adj_mproduct_merge: TN=%lld is not T*N", (long long)TN);
  if (lo > T - 1) lo = T - 1;
  if (hi > T - 1) hi = T - 1;
  const int W = lo + hi + 1;
  TMGCN_REQUIRE(W <= 64, "adj_mproduct_merge: band of %d slices is wider than 64 (use tmgcn_adj_mproduct_expand)", W);
  if (TN == 0) return TMGCN_OK;
  TMGCN_REQUIRE(rowptr && M && ldm >= T, "adj_mproduct_merge: null pointer / bad M");
  if (W <= 32)
    hipLaunchKernelGGL((adj_mproduct_merge_kernel<32, FILL>), dim3(blocks(TN * 32)), dim3(256), 0, st, rowptr, col, val, TN,
                       N, T, M, ldm, lo, hi, out_count, out_rowptr, out_col, out_val);
  else
    hipLaunchKernelGGL((adj_mproduct_merge_kernel<64, FILL>), dim3(blocks(TN * 64)), dim3(256), 0, st, rowptr, col, val, TN,
                       N, T, M, ldm, lo, hi, out_count, out_rowptr, out_col, out_val);
  return check_launch("adj_mproduct_merge");
}

extern "C" int tmgcn_adj_mproduct_merge_count(const int64_t* rowptr, const int32_t* col, int64_t TN, int32_t N, int32_t T,
                                               const float* M, int32_t ldm, int32_t band_lo, int32_t band_hi,
                                               int64_t* out_count, void* stream) {
  TMGCN_REQUIRE(out_count, "adj_mproduct_merge_count: null output");
  if (TN == 0) {
    (void)hipMemsetAsync(out_count, 0, sizeof(int64_t), (hipStream_t)stream);
    return check_launch("adj_mproduct_merge_count memset");
  }
  return mproduct_merge_launch<false>(rowptr, col, nullptr, TN, N, T, M, ldm, band_lo, band_hi, out_count, nullptr, nullptr,
                                      nullptr, (hipStream_t)stream);
}

extern "C" int tmgcn_adj_mproduct_merge_fill(const int64_t* rowptr, const int32_t* col, const float* val, int64_t TN,
                                              int32_t N, int32_t T, const float* M, int32_t ldm, int32_t band_lo,
                                              int32_t band_hi, const int64_t* out_rowptr, int32_t* out_col, float* out_val,
                                              void* stream) {
  // col / val / out_col / out_val may be null when the tensor has no stored entry at all (never dereferenced then)
  TMGCN_REQUIRE(TN == 0 || out_rowptr, "adj_mproduct_merge_fill: null out_rowptr");
  return mproduct_merge_launch<true>(rowptr, col, val, TN, N, T, M, ldm, band_lo, band_hi, nullptr, out_rowptr, out_col,
                                     out_val, (hipStream_t)stream);
}

// sorted keys -> CSR arrays
extern "C" int tmgcn_adj_keys_to_csr(const uint64_t* key, int64_t n, int64_t N, int64_t TN, int64_t* rowptr,
                                      int32_t* col, void* stream) {
  TMGCN_REQUIRE(n >= 0 && N > 0 && TN >= 0, "adj_keys_to_csr: bad size");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adj_rowptr_kernel, dim3(blocks(n ? n : TN + 1)), dim3(256), 0, st, key, n, N, TN, rowptr);
  if (n) hipLaunchKernelGGL(adj_cols_kernel, dim3(blocks(n)), dim3(256), 0, st, key, n, N, col);
  return check_launch("adj_keys_to_csr");
}

// keys of the per-slice transpose of a batched CSR (then sort them with tmgcn_coo_sort_reduce)
extern "C" int tmgcn_adj_transpose_keys(const int64_t* rowptr, const int32_t* col, int64_t TN, int64_t N,
                                         uint64_t* okey, void* stream) {
  TMGCN_REQUIRE(TN >= 0 && N > 0, "adj_transpose_keys: bad size");
  if (TN == 0) return TMGCN_OK;
  hipLaunchKernelGGL(adj_transpose_keys_kernel, dim3(blocks(TN)), dim3(256), 0, (hipStream_t)stream, rowptr, col,
                     TN, N, okey);
  return check_launch("adj_transpose_keys");
}

/*
 * A caller of the C-ABI that knows nothing about torch or Python: plain C99 + the HIP runtime.
 * It runs one TM-GCN layer forward and backward through include/tmgcn.h on small seeded inputs
 *
 *     Xt = M x1 X          tmgcn_mtransform_f32            (ehf:204)
 *     AX = A_k Xt_k        tmgcn_spmm_csr_batched_f32      (ehf:206-207)
 *     Y  = AX W            tmgcn_gemm_f32                  (ehf:222)
 *     Y' = (A * Xt) W      tmgcn_spmm_gemm_f32, fused      (same statements, one launch)
 *     dW = AX^T dY         tmgcn_gemm_dw_f32               (autograd of ehf:222)
 *     dA = dY W^T          tmgcn_gemm_f32, trans_w = 1
 *     dX = M^T x1 (A^T dA) transposed CSR built on the host here
 *
 * and compares every result with the plain-C oracle (oracle/tmgcn_ref.c, linked as a checker).
 * Exit status 0 and a line "abi_driver OK max_rel_err=..." on success.  Built and run by
 * tests/test_gpu_abi_driver.py:
 *     gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include driver.c -Iinclude -Ltm-gcn_amd -ltmgcn_hip
 *         -Loracle -ltmgcn_ref -L/opt/rocm/lib -lamdhip64 -lm
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tmgcn.h"

/* oracle/tmgcn_ref.c (test infrastructure) */
void ref_mtransform(const double* M, int T, int transpose, const float* X, float* Y, int64_t C);
void ref_spmm(const int64_t* rowptr, const int32_t* col, const float* val, const float* X, float* Y,
              int64_t n_rows, int32_t N, int32_t F);
void ref_gemm(const float* A, const float* W, float* Y, int64_t R, int32_t K, int32_t Nf, int32_t trans_w,
              int64_t rows_per_batch, int64_t w_batch_stride);
void ref_gemm_dw(const float* A, const float* dY, float* dW, int64_t R, int32_t K, int32_t Nf,
                 int64_t rows_per_batch);

#define HIP(x)                                                                        \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d HIP error %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)
#define ABI(x)                                                                            \
  do {                                                                                    \
    int s_ = (x);                                                                         \
    if (s_ != 0) {                                                                        \
      fprintf(stderr, "%s:%d ABI status %d: %s\n", __FILE__, __LINE__, s_, tmgcn_last_error()); \
      exit(3);                                                                            \
    }                                                                                     \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { /* xorshift64* */
  rng_state ^= rng_state >> 12;
  rng_state ^= rng_state << 25;
  rng_state ^= rng_state >> 27;
  return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}
static float unif(void) { return (float)(rnd() >> 8) / 16777216.0f - 0.5f; }

static void* to_dev(const void* h, size_t bytes) {
  void* d = NULL;
  HIP(hipMalloc(&d, bytes ? bytes : 4));
  if (bytes) HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
  return d;
}
static float* from_dev(const void* d, size_t n) {
  float* h = (float*)malloc(n * sizeof(float));
  HIP(hipMemcpy(h, d, n * sizeof(float), hipMemcpyDeviceToHost));
  return h;
}
static double worst = 0.0;
static void compare(const char* what, const float* got, const float* ref, size_t n) {
  double mx = 0.0, scale = 1e-30;
  for (size_t i = 0; i < n; ++i) {
    const double d = fabs((double)got[i] - (double)ref[i]);
    if (d > mx) mx = d;
    if (fabs((double)ref[i]) > scale) scale = fabs((double)ref[i]);
  }
  const double rel = mx / scale;
  printf("  %-28s max|d|/max|ref| = %.3e\n", what, rel);
  if (rel > worst) worst = rel;
  if (!(rel <= 1e-5)) {
    fprintf(stderr, "abi_driver FAILED: %s off by %.3e\n", what, rel);
    exit(1);
  }
}

int main(void) {
  enum { T = 6, N = 700, F = 32, NF = 16, BAND = 3 };
  const int64_t R = (int64_t)T * N;
  printf("tmgcn ABI version %d\n", tmgcn_abi_version());

  /* band M (1/(d+1) on BAND lower diagonals), fp32 for the device, fp64 for the oracle */
  float Mf[T * T];
  double Md[T * T];
  memset(Mf, 0, sizeof(Mf));
  for (int k = 0; k < T; ++k)
    for (int d = 0; d < BAND && d <= k; ++d) Mf[k * T + (k - d)] = 1.0f / (float)(d + 1);
  for (int i = 0; i < T * T; ++i) Md[i] = Mf[i];

  /* batched CSR: ragged rows (0..9 non-zeros, some rows empty), unsorted draws sorted per row */
  int64_t* rowptr = (int64_t*)malloc((R + 1) * sizeof(int64_t));
  rowptr[0] = 0;
  for (int64_t r = 0; r < R; ++r) rowptr[r + 1] = rowptr[r] + (rnd() % 10 == 0 ? 0 : rnd() % 10);
  const int64_t nnz = rowptr[R];
  int32_t* col = (int32_t*)malloc(nnz * sizeof(int32_t));
  float* val = (float*)malloc(nnz * sizeof(float));
  for (int64_t p = 0; p < nnz; ++p) {
    col[p] = (int32_t)(rnd() % N);
    val[p] = unif();
  }
  /* transposed CSR per slice (counting sort by column) for the backward SpMM */
  int64_t* trowptr = (int64_t*)calloc(R + 1, sizeof(int64_t));
  int32_t* tcol = (int32_t*)malloc(nnz * sizeof(int32_t));
  float* tval = (float*)malloc(nnz * sizeof(float));
  for (int64_t r = 0; r < R; ++r)
    for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) trowptr[(r / N) * N + col[p] + 1]++;
  for (int64_t r = 0; r < R; ++r) trowptr[r + 1] += trowptr[r];
  {
    int64_t* fill = (int64_t*)malloc(R * sizeof(int64_t));
    memcpy(fill, trowptr, R * sizeof(int64_t));
    for (int64_t r = 0; r < R; ++r)
      for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
        const int64_t q = fill[(r / N) * N + col[p]]++;
        tcol[q] = (int32_t)(r % N);
        tval[q] = val[p];
      }
    free(fill);
  }

  float* X = (float*)malloc(R * F * sizeof(float));
  float* W = (float*)malloc(F * NF * sizeof(float));
  float* dY = (float*)malloc(R * NF * sizeof(float));
  for (int64_t i = 0; i < R * F; ++i) X[i] = unif();
  for (int i = 0; i < F * NF; ++i) W[i] = unif();
  for (int64_t i = 0; i < R * NF; ++i) dY[i] = unif();

  /* ---- oracle ---- */
  float* Xt_r = (float*)malloc(R * F * sizeof(float));
  float* AX_r = (float*)malloc(R * F * sizeof(float));
  float* Y_r = (float*)malloc(R * NF * sizeof(float));
  float* dW_r = (float*)malloc(F * NF * sizeof(float));
  float* dA_r = (float*)malloc(R * F * sizeof(float));
  float* dXt_r = (float*)malloc(R * F * sizeof(float));
  float* dX_r = (float*)malloc(R * F * sizeof(float));
  ref_mtransform(Md, T, 0, X, Xt_r, (int64_t)N * F);
  ref_spmm(rowptr, col, val, Xt_r, AX_r, R, N, F);
  ref_gemm(AX_r, W, Y_r, R, F, NF, 0, 0, 0);
  ref_gemm_dw(AX_r, dY, dW_r, R, F, NF, 0);
  ref_gemm(dY, W, dA_r, R, NF, F, 1, 0, 0);
  ref_spmm(trowptr, tcol, tval, dA_r, dXt_r, R, N, F);
  ref_mtransform(Md, T, 1, dXt_r, dX_r, (int64_t)N * F);

  /* ---- device, through the C-ABI ---- */
  hipStream_t st;
  HIP(hipStreamCreate(&st));
  float* dM = (float*)to_dev(Mf, sizeof(Mf));
  int64_t* d_rowptr = (int64_t*)to_dev(rowptr, (R + 1) * sizeof(int64_t));
  int32_t* d_col = (int32_t*)to_dev(col, nnz * sizeof(int32_t));
  float* d_val = (float*)to_dev(val, nnz * sizeof(float));
  int64_t* d_trowptr = (int64_t*)to_dev(trowptr, (R + 1) * sizeof(int64_t));
  int32_t* d_tcol = (int32_t*)to_dev(tcol, nnz * sizeof(int32_t));
  float* d_tval = (float*)to_dev(tval, nnz * sizeof(float));
  float* dX_in = (float*)to_dev(X, R * F * sizeof(float));
  float* dWt = (float*)to_dev(W, F * NF * sizeof(float));
  float* d_dY = (float*)to_dev(dY, R * NF * sizeof(float));
  float *dXt, *dAX, *dYo, *dYf, *dAXf, *d_dW, *d_dA, *d_dXt, *d_dXt2, *d_dX;
  HIP(hipMalloc((void**)&dXt, R * F * 4));
  HIP(hipMalloc((void**)&dAX, R * F * 4));
  HIP(hipMalloc((void**)&dYo, R * NF * 4));
  HIP(hipMalloc((void**)&dYf, R * NF * 4));
  HIP(hipMalloc((void**)&dAXf, R * F * 4));
  HIP(hipMalloc((void**)&d_dW, F * NF * 4));
  HIP(hipMalloc((void**)&d_dA, R * F * 4));
  HIP(hipMalloc((void**)&d_dXt, R * F * 4));
  HIP(hipMalloc((void**)&d_dXt2, R * F * 4));
  HIP(hipMalloc((void**)&d_dX, R * F * 4));
  const int64_t ws_bytes = tmgcn_gemm_dw_workspace_bytes(R, F, NF, 0);
  void* ws = NULL;
  HIP(hipMalloc(&ws, ws_bytes > 0 ? (size_t)ws_bytes : 4));

  /* forward */
  ABI(tmgcn_mtransform_f32(dM, T, T, 0, 0, 0, T, T, BAND - 1, 0, dX_in, dXt, (int64_t)N * F, 0, 0, st));
  ABI(tmgcn_spmm_csr_batched_f32(d_rowptr, d_col, d_val, dXt, dAX, R, N, F, st));
  ABI(tmgcn_gemm_f32(dAX, dWt, dYo, NULL, R, F, NF, 0, 0, 0, TMGCN_ACT_NONE, TMGCN_GEMM_AUTO, st));
  if (!tmgcn_spmm_gemm_supported(F, NF)) {
    fprintf(stderr, "fused kernel should support K=%d Nf=%d\n", F, NF);
    return 1;
  }
  ABI(tmgcn_spmm_gemm_f32(d_rowptr, d_col, d_val, dXt, R, N, F, dWt, NF, 0, 0, 0, TMGCN_ACT_NONE, dYf, dAXf, NULL, 0, st));
  /* backward */
  ABI(tmgcn_gemm_dw_f32(dAX, d_dY, d_dW, R, F, NF, 0, TMGCN_DW_AUTO, ws, ws_bytes, st));
  ABI(tmgcn_gemm_f32(d_dY, dWt, d_dA, NULL, R, NF, F, 1, 0, 0, TMGCN_ACT_NONE, TMGCN_GEMM_AUTO, st));
  ABI(tmgcn_spmm_csr_batched_f32(d_trowptr, d_tcol, d_tval, d_dA, d_dXt, R, N, F, st));
  ABI(tmgcn_mtransform_f32(dM, T, T, 1, 0, 0, T, T, 0, BAND - 1, d_dXt, d_dX, (int64_t)N * F, 0, 0, st));
  /* the backward pair as ONE fused launch: A^T (dY W^T) = (A^T dY) W^T */
  if (tmgcn_spmm_gemm_supported(NF, F))
    ABI(tmgcn_spmm_gemm_f32(d_trowptr, d_tcol, d_tval, d_dY, R, N, NF, dWt, F, 1, 0, 0, TMGCN_ACT_NONE, d_dXt2, NULL, NULL, 0, st));
  /* ABI 3: the column-window form — P1 in two unequal column chunks written into a second buffer must
   * reproduce the one-shot result (the consumer of the node-chunked all-gather) */
  float* dXt_w = NULL;
  HIP(hipMalloc((void**)&dXt_w, R * F * 4));
  HIP(hipMemsetAsync(dXt_w, 0xff, R * F * 4, st));
  {
    const int64_t C = (int64_t)N * F, c_split = (int64_t)(N / 3) * F;
    ABI(tmgcn_mtransform_ld_f32(dM, T, T, 0, 0, 0, T, T, BAND - 1, 0, dX_in, C, dXt_w, C, c_split, 0, 0, st));
    ABI(tmgcn_mtransform_ld_f32(dM, T, T, 0, 0, 0, T, T, BAND - 1, 0, dX_in + c_split, C, dXt_w + c_split, C, C - c_split, 0, 0, st));
  }
  HIP(hipStreamSynchronize(st));

  compare("M-transform", from_dev(dXt, R * F), Xt_r, R * F);
  {
    float* whole = from_dev(dXt, R * F);
    float* win = from_dev(dXt_w, R * F);
    if (memcmp(whole, win, (size_t)R * F * 4) != 0) {
      fprintf(stderr, "column-window M-transform differs from the one-shot product\n");
      return 1;
    }
    printf("%-28s bit-equal to the one-shot product\n", "M-transform, column windows");
  }
  compare("batched CSR SpMM", from_dev(dAX, R * F), AX_r, R * F);
  compare("GEMM", from_dev(dYo, R * NF), Y_r, R * NF);
  compare("fused SpMM+GEMM: Y", from_dev(dYf, R * NF), Y_r, R * NF);
  compare("fused SpMM+GEMM: AX", from_dev(dAXf, R * F), AX_r, R * F);
  compare("dW", from_dev(d_dW, F * NF), dW_r, F * NF);
  compare("dA = dY W^T", from_dev(d_dA, R * F), dA_r, R * F);
  compare("transposed SpMM", from_dev(d_dXt, R * F), dXt_r, R * F);
  compare("M^T-transform (dX)", from_dev(d_dX, R * F), dX_r, R * F);
  if (tmgcn_spmm_gemm_supported(NF, F)) compare("fused backward pair", from_dev(d_dXt2, R * F), dXt_r, R * F);

  /* error contract: negative status + message, nothing launched */
  if (tmgcn_spmm_csr_batched_f32(NULL, d_col, d_val, dXt, dAX, R, N, F, st) >= 0 || !tmgcn_last_error()[0]) {
    fprintf(stderr, "null rowptr must be rejected with a message\n");
    return 1;
  }
  printf("abi_driver OK max_rel_err=%.3e nnz=%lld\n", worst, (long long)nnz);
  return 0;
}

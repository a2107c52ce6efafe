/* CPU ORACLE (plain C) for the TM-GCN layer kernels — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * A loop-level restatement, independent of PyTorch, of the arithmetic the reference performs on
 * its hot path (/root/reference/TensorGCN-master/embedding_help_functions.py, "ehf"): fp64
 * accumulation rounded once into fp32 outputs, as the reference's fp64 matmul / sparse.mm
 * written into `t.zeros(...)` fp32 buffers do (ehf:204-207).  Uses the same batched-CSR layout
 * as include/tmgcn.h so the HIP kernels can be compared call for call.
 *
 * Parity status: PINNED through tests/test_oracle_golden.py (checked against fixtures captured
 * from the real ehf, tests/golden/make_golden.py): ref_mtransform + ref_spmm against G1's AtXt,
 * ref_gemm + ref_gemm_dw against G2's logits and dW (shared and per-slice W), ref_mtransform_rows
 * against ref_mtransform.
 */
#include <stdint.h>
#include <stddef.h>

/* P1, ehf:204 — Y[k][c] = sum_j Mop[k][j] X[j][c]; Mop = M or M^T (autograd backward). */
void ref_mtransform(const double* M, int T, int transpose, const float* X, float* Y, int64_t C) {
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < C; ++c) {
    for (int k = 0; k < T; ++k) {
      double s = 0.0;
      for (int j = 0; j < T; ++j) {
        const double m = transpose ? M[(size_t)j * T + k] : M[(size_t)k * T + j];
        if (m != 0.0) s += m * (double)X[(size_t)j * C + c];
      }
      Y[(size_t)k * C + c] = (float)s;
    }
  }
}

/* The same statement for a WINDOW of output rows [row0, row0 + nrows) only (bench.py's verify leg
 * needs one output slice of fibres that span all T input slices: computing all T rows would be
 * T times the work).  Y is [nrows][C]. */
void ref_mtransform_rows(const double* M, int T, int transpose, int row0, int nrows, const float* X, float* Y,
                         int64_t C) {
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < C; ++c) {
    for (int k = row0; k < row0 + nrows; ++k) {
      double s = 0.0;
      for (int j = 0; j < T; ++j) {
        const double m = transpose ? M[(size_t)j * T + k] : M[(size_t)k * T + j];
        if (m != 0.0) s += m * (double)X[(size_t)j * C + c];
      }
      Y[(size_t)(k - row0) * C + c] = (float)s;
    }
  }
}

/* P2, ehf:206-207 — Y[r][f] = sum_p val[p] X[(r/N)*N + col[p]][f] over the batched CSR. */
void ref_spmm(const int64_t* rowptr, const int32_t* col, const float* val, const float* X,
              float* Y, int64_t n_rows, int32_t N, int32_t F) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < n_rows; ++r) {
    const int64_t base = (r / N) * (int64_t)N;
    for (int f = 0; f < F; ++f) {
      double s = 0.0;
      for (int64_t p = rowptr[r]; p < rowptr[r + 1]; ++p)
        s += (double)val[p] * (double)X[(size_t)(base + col[p]) * F + f];
      Y[(size_t)r * F + f] = (float)s;
    }
  }
}

/* P3, ehf:222 — Y[r][n] = sum_k A[r][k] Wb[k][n] (trans_w: Wb[n][k]); rows_per_batch = 0: shared W. */
void ref_gemm(const float* A, const float* W, float* Y, int64_t R, int32_t K, int32_t Nf,
              int32_t trans_w, int64_t rows_per_batch, int64_t w_batch_stride) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const float* Wb = W + (rows_per_batch ? (r / rows_per_batch) * w_batch_stride : 0);
    for (int n = 0; n < Nf; ++n) {
      double s = 0.0;
      for (int k = 0; k < K; ++k)
        s += (double)A[(size_t)r * K + k] *
             (double)(trans_w ? Wb[(size_t)n * K + k] : Wb[(size_t)k * Nf + n]);
      Y[(size_t)r * Nf + n] = (float)s;
    }
  }
}

/* autograd of P3 w.r.t. W — dW_b[k][n] = sum_{r in batch b} A[r][k] dY[r][n]. */
void ref_gemm_dw(const float* A, const float* dY, float* dW, int64_t R, int32_t K, int32_t Nf,
                 int64_t rows_per_batch) {
  const int64_t br = rows_per_batch ? rows_per_batch : R;
  const int64_t nb = br ? (R + br - 1) / br : 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t b = 0; b < nb; ++b) {
    for (int k = 0; k < K; ++k) {
      for (int n = 0; n < Nf; ++n) {
        double s = 0.0;
        const int64_t r1 = (b + 1) * br < R ? (b + 1) * br : R;
        for (int64_t r = b * br; r < r1; ++r)
          s += (double)A[(size_t)r * K + k] * (double)dY[(size_t)r * Nf + n];
        dW[((size_t)b * K + k) * Nf + n] = (float)s;
      }
    }
  }
}

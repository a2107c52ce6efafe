/*
 * tmgcn.h — C-ABI of the MI355X (gfx950) TM-GCN propagation layer.
 *
 * This is the drop-in boundary for the ONE hot path of IBM/TM-GCN: the tensor
 * M-product graph-convolution layer  Y = (Â ⋆_M X) W  forward and backward.
 * The reference has no FFI of its own (it is flat PyTorch-CPU Python); every
 * entry point below names the reference statement(s) it replaces, relative to
 * TensorGCN-master/embedding_help_functions.py ("ehf").  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (hipMalloc / torch ROCm storage) unless
 *     the parameter name ends in _host.  No torch types cross this boundary.
 *   - All tensors are dense row-major with the innermost dimension contiguous.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every
 *     launcher is asynchronous on that stream, allocates nothing and never
 *     synchronises, so a caller may capture it into a hipGraph.
 *   - Return value: 0 on success, a negative tmgcn_status otherwise;
 *     tmgcn_last_error() returns a thread-local message for the last failure.
 *
 * Data layout in HBM
 *   feature tensor  X[T][N][F]  fp32, tube fibre stride N*F, viewed as [T][C=N*F] by P1
 *                               and as [R=T*N][F] by P2/P3.
 *   batched CSR     rowptr[T*N+1] int64 (global offsets into col/val; row r = k*N+i),
 *                   col[nnz] int32 (column inside the slice, 0..N-1), val[nnz] fp32.
 *                   Slice k is rows [k*N,(k+1)*N).  Total nnz may exceed 2^31.
 */
#ifndef TMGCN_H
#define TMGCN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  TMGCN_OK = 0,
  TMGCN_ERR_INVALID = -1,  /* bad argument (shape, null pointer, unsupported size) */
  TMGCN_ERR_LAUNCH = -2,   /* HIP reported an error at launch                       */
  TMGCN_ERR_WORKSPACE = -3 /* caller-provided workspace too small                   */
} tmgcn_status;

/* activation ids for the fused P3 epilogue / P5 pointwise (ehf:284-289, 455-460) */
enum { TMGCN_ACT_NONE = 0, TMGCN_ACT_RELU = 1, TMGCN_ACT_LEAKY = 2, TMGCN_ACT_SELU = 3 };

/* ABI version 5 = version 4 + tmgcn_pool_stats + the row_blocks partition argument of tmgcn_layer12_fwd/bwd_f32, the backward's AX / dW2 pair and tmgcn_layer12_bwd_forms_dw2 + tmgcn_head_loss_combine_f32 / tmgcn_head_loss_lanes (split rows of the one-pass head + loss plan) + the giant-row
 *   plan entry points tmgcn_spmm_csr_batched_f32_plan / tmgcn_spmm_gemm_f32_plan / tmgcn_spmm_giant_workspace_bytes; the launchers' scratch words (tile counters, hand-off blocks) are kept apart
 *   per stream (eager launches) and per recorded launch (hipGraph capture), and a launcher that cannot keep two launches
 *   apart returns TMGCN_ERR_INVALID with the reason instead of re-using a word that may be in flight (csrc/pools.hip).
 * ABI version 4 = version 3 + tmgcn_head_loss_f32 / tmgcn_scale2_f32 (one-pass edge head + loss + gradients).
 * ABI version 3 = version 2 + tmgcn_mtransform_ld_f32 (column-window M-transform) + tmgcn_adj_mproduct_merge_* (segmented-merge M-product).
 * ABI version 2: no process-wide settings.  What used to be tmgcn_config_set() knobs are
 * per-call arguments (grid_reserve of tmgcn_spmm_gemm_f32, algo of tmgcn_gemm_dw_f32): two callers
 * in one process never see each other's choices. */
int tmgcn_abi_version(void);
const char* tmgcn_last_error(void);
/* State of the library's scratch words on the current device (synchronises the device; diagnostics and tests):
 *   out[0] streams that hold an eager slot (at most 64)      out[1] / out[4] tile counters handed to recorded launches / capacity
 *   out[2] / out[5] hand-off blocks handed to recorded launches / capacity
 *   out[3] int32 words of the hand-off blocks that are NOT zero while the device is idle (must be 0: every launch
 *          leaves its block zero; anything else means a kernel died mid-flight or two launches shared a block) */
int tmgcn_pool_stats(int64_t* out, int32_t n);

/* algorithm of tmgcn_gemm_f32 (instruction choice only; both fp32-accurate and reproducible) */
enum {
  TMGCN_GEMM_AUTO = 0,    /* bf16 matrix cores after an exact 3-way split of the fp32 operands where the shapes
                             allow (K a multiple of 4 in [16, 512] — above 128 as 128-wide k-chunks —, 16-byte aligned A), else exact f32 */
  TMGCN_GEMM_F32MFMA = 1  /* always the exact-f32 MFMA kernel: bitwise an fmaf chain in k order */
};

/* algorithm of tmgcn_gemm_dw_f32 (performance / instruction choice only; both are fp32-accurate,
 * atomic-free and bitwise reproducible) */
enum {
  TMGCN_DW_AUTO = 0,    /* bf16 matrix cores after an exact 3-way split of the fp32 operands where the
                           shapes allow (K, Nf >= 16, multiples of 4, 16-byte aligned), else exact f32 */
  TMGCN_DW_F32MFMA = 1  /* always the exact-f32 MFMA kernel (v_mfma_f32_32x32x2_f32) */
};

/* ---- P1: tube-fibre M-transform ------------------------------------------------
 * Replaces  t.matmul(self.M, X.reshape(self.T,-1)).reshape(X.size())
 *   ehf:204, ehf:308, ehf:404 (forward), ehf:346 (third application),
 *   ehf:224/332/341 (Minv), and autograd's Mᵀ product in backward.
 *
 *   Y[k][c] = sum_{j=0..T_in-1} Mop[row_off + k][col_off + j] * X[j][c]
 *        k = 0..T_out-1, c = 0..C-1,   Mop = M (transpose=0) or Mᵀ (transpose=1)
 *
 * M is a dense [Tm][Tm] fp32 matrix with leading dimension ldm.  The caller
 * promises Mop[a][b] == 0 unless  a - band_lo <= b <= a + band_hi  (a, b are
 * indices into Mop, offsets included); pass band_lo = band_hi = Tm for a dense M.
 * Entries outside the promised band are never read.  X and Y must not overlap.
 *
 * x_group_rows / y_group_rows (0 = plain row order): group-interleaved row storage used as
 * the send / receive layout of the slice<->node all-to-all of the sharded layer.  With
 * g > 0, logical row k of a T-row tensor is stored at row (k % g) * (T / g) + k / g, i.e. as
 * [g][T/g][C]: block kk holds row kk of every rank's slice range, contiguous per destination.
 */
int tmgcn_mtransform_f32(const float* M, int32_t Tm, int32_t ldm, int32_t transpose,
                         int32_t row_off, int32_t col_off, int32_t T_out, int32_t T_in,
                         int32_t band_lo, int32_t band_hi,
                         const float* X, float* Y, int64_t C,
                         int32_t x_group_rows, int32_t y_group_rows, void* stream);
/* The same on a COLUMN WINDOW of wider tensors (ABI 3): X rows are ldx floats apart, Y rows ldy
 * floats apart (ldx, ldy >= C; X / Y point at the window's first column).  This is the consumer
 * of the node-chunked all-gather of the slice-sharded layer: a gathered chunk [T][Nc*F] is
 * transformed straight into columns [c0*F, (c0+Nc)*F) of the resident [T/G][N*F] result (and, in
 * backward, the window of the upstream gradient straight into the reduce-scatter send buffer), so
 * the replicated [T][N*F] tensor of ehf:204 / 308 never exists.  Per output element the
 * arithmetic is that of tmgcn_mtransform_f32 (same kernel): chunked == unchunked, bit for bit. */
int tmgcn_mtransform_ld_f32(const float* M, int32_t Tm, int32_t ldm, int32_t transpose,
                            int32_t row_off, int32_t col_off, int32_t T_out, int32_t T_in,
                            int32_t band_lo, int32_t band_hi,
                            const float* X, int64_t ldx, float* Y, int64_t ldy, int64_t C,
                            int32_t x_group_rows, int32_t y_group_rows, void* stream);

/* ---- P2: batched CSR SpMM (per-frontal-slice Â_k · X_k) -------------------------
 * Replaces the loops  for k in range(T): AtXt[k] = t.sparse.mm(At[k], Xt[k])
 *   ehf:206-207, 303-304, 310-311, 406-407, 471-472; with the transposed CSR it is
 *   autograd's sparse.mm backward  dXt[k] = Â_kᵀ · dAtXt[k].
 *
 *   Y[r][f] = sum_{p=rowptr[r]}^{rowptr[r+1]-1} val[p] * X[(r / N) * N + col[p]][f]
 *        r = 0..n_rows-1 (n_rows = T*N), f = 0..F-1
 *
 * Row sums are formed in a fixed order (no atomics): bitwise reproducible.
 */
/* Giant rows (optional plan of the two SpMM launchers below; csr.BatchedCSR.giant_plan builds it): a row of more than
 * TMGCN_GIANT_ROW stored entries is cut into chunks of TMGCN_GIANT_CHUNK entries (the last one shorter), each summed by one
 * block of a small launch in front of the main kernel, which then adds the partial sums in chunk order instead of gathering the
 * row on the four waves of ONE block (41 M entries/s: a 10^6-entry row would cost 23 ms whatever else the launch holds).
 *   giant_rows    int64 [n_giant]      ascending global row indices (k*N + i) of EVERY row with more than TMGCN_GIANT_ROW entries
 *   giant_chunks  int32 [n_giant + 1 + n_giant_chunks]  first chunk of each giant row (n_giant + 1 values, 0 … n_giant_chunks),
 *                                      then for every chunk the index of its giant row
 *   giant_ws      float [n_giant_chunks][F]   workspace (tmgcn_spmm_giant_workspace_bytes)
 * All NULL / 0 = no plan.  Results are bit-reproducible for a given plan; with and without a plan they differ by fp32
 * summation order on the giant rows only. */
#define TMGCN_GIANT_ROW 32768
#define TMGCN_GIANT_CHUNK 4096
int64_t tmgcn_spmm_giant_workspace_bytes(int32_t n_giant_chunks, int32_t F);
int tmgcn_spmm_csr_batched_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                               const float* X, float* Y, int64_t n_rows, int32_t N, int32_t F,
                               void* stream);
/* Same, with the caller's average stored non-zeros per row (< 0: unknown); it only steers
 * how many lanes share a row in the small-F kernel, i.e. the order in which a row's terms are
 * added (results agree to fp32 rounding; for a given hint they are bitwise reproducible). */
int tmgcn_spmm_csr_batched_f32_hint(const int64_t* rowptr, const int32_t* col, const float* val,
                                    const float* X, float* Y, int64_t n_rows, int32_t N,
                                    int32_t F, float avg_nnz_per_row, void* stream);
/* … with a giant-row plan (above; used by the F % 4 == 0, F >= 16 kernel, ignored by the narrow ones) */
int tmgcn_spmm_csr_batched_f32_plan(const int64_t* rowptr, const int32_t* col, const float* val,
                                    const float* X, float* Y, int64_t n_rows, int32_t N,
                                    int32_t F, float avg_nnz_per_row, const int64_t* giant_rows,
                                    const int32_t* giant_chunks, int32_t n_giant, int32_t n_giant_chunks,
                                    float* giant_ws, int64_t giant_ws_bytes, void* stream);

/* ---- P2+P3 fused: Y = act((Â ⋆ X) · Wop) in one launch ------------------------------
 * Replaces the pair  sparse.mm loop + t.matmul(AtXt, W)  (ehf:206-207 + 222, 303-304 + 349,
 * 471-472 + 486-489).  With the transposed CSR and trans_w=1 it is the backward pair, using
 * Âᵀ(dY·Wᵀ) = (Âᵀ·dY)·Wᵀ.  X is [n_rows][K]; W, trans_w, rows_per_batch, w_batch_stride, act,
 * pre_act as in tmgcn_gemm_f32.  AX (optional, may be NULL) receives the SpMM result
 * [n_rows][K] itself (needed for dW).  grid_reserve (>= 0, performance only, never results): block
 * slots of the persistent grid THIS launch leaves free — the sharded layer passes one per CU so that
 * RCCL's exchange kernels on a side stream can become resident next to the compute kernel.  Supported when tmgcn_spmm_gemm_supported(K, Nf) != 0
 * (K a multiple of 8 in [16, 128] with Nf <= 128 — MFMA epilogue — or K in {1,2,3,4,6,8} with Nf <= 16 —
 * the reference's real widths, FMA epilogue); otherwise call the two kernels separately.
 * The MFMA form reads X rows and writes AX rows as 16-byte vectors: X and AX must be 16-byte aligned
 * (TMGCN_ERR_INVALID otherwise; the torch.ops layer copies a misaligned X once instead).
 */
int tmgcn_spmm_gemm_supported(int32_t K, int32_t Nf);
int tmgcn_spmm_gemm_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                        const float* X, int64_t n_rows, int32_t N, int32_t K,
                        const float* W, int32_t Nf, int32_t trans_w,
                        int64_t rows_per_batch, int64_t w_batch_stride, int32_t act,
                        float* Y, float* AX, float* pre_act, int32_t grid_reserve, void* stream);
/* Same, with the caller's average stored non-zeros per row (< 0: unknown) — steers how many lanes
 * share a row in the narrow (K <= 8) kernel, i.e. the order in which a row's terms are added, as in
 * tmgcn_spmm_csr_batched_f32_hint; and, for K = 64 / 128, how the products are formed: exact-f32 MFMAs (an fmaf chain in
 * k order; always without a hint), or — below 14 non-zeros per row, where the products are as long as the gather — on the
 * bf16 matrix cores after an exact 3-way split of both operands, i.e. the numerics of tmgcn_gemm_f32's TMGCN_GEMM_AUTO
 * (the unfused default route).  AX does not depend on the hint. */
int tmgcn_spmm_gemm_f32_hint(const int64_t* rowptr, const int32_t* col, const float* val,
                             const float* X, int64_t n_rows, int32_t N, int32_t K,
                             const float* W, int32_t Nf, int32_t trans_w,
                             int64_t rows_per_batch, int64_t w_batch_stride, int32_t act,
                             float* Y, float* AX, float* pre_act, int32_t grid_reserve,
                             float avg_nnz_per_row, void* stream);
/* … with a giant-row plan (the MFMA-epilogue kernel, K a multiple of 8 in [16, 128]; ignored by the narrow one) */
int tmgcn_spmm_gemm_f32_plan(const int64_t* rowptr, const int32_t* col, const float* val,
                             const float* X, int64_t n_rows, int32_t N, int32_t K,
                             const float* W, int32_t Nf, int32_t trans_w,
                             int64_t rows_per_batch, int64_t w_batch_stride, int32_t act,
                             float* Y, float* AX, float* pre_act, int32_t grid_reserve,
                             float avg_nnz_per_row, const int64_t* giant_rows, const int32_t* giant_chunks,
                             int32_t n_giant, int32_t n_giant_chunks, float* giant_ws, int64_t giant_ws_bytes,
                             void* stream);

/* ---- P3: feature·weight contraction ----------------------------------------------
 * Replaces  t.matmul(AtXt, Wt)  ehf:222, 330, 340, 344, 349, 415, 486-489.
 *
 *   Y[r][n] = act( sum_k A[r][k] * Wop_b[k][n] ),   b = rows_per_batch ? r / rows_per_batch : 0
 *   Wop_b = W_b (trans_w=0, W_b is [K][Nf]) or W_bᵀ (trans_w=1, W_b is [Nf][K]);
 *   W_b = W + b * w_batch_stride.  rows_per_batch = 0 is the reference's condensed_W
 *   (one shared weight, ehf:189); rows_per_batch = N gives one weight per slice (ehf:191).
 *   trans_w=1 is the backward  dA = dY · Wᵀ.
 *   pre_act (optional, may be NULL): when act != NONE also store the pre-activation sum.
 *   algo: TMGCN_GEMM_AUTO / TMGCN_GEMM_F32MFMA (per call).
 */
int tmgcn_gemm_f32(const float* A, const float* W, float* Y, float* pre_act,
                   int64_t R, int32_t K, int32_t Nf, int32_t trans_w,
                   int64_t rows_per_batch, int64_t w_batch_stride, int32_t act, int32_t algo, void* stream);

/* The same contraction with the weight STORED in bf16 (the "bf16 weights" configuration of the
 * AMLSim experiment, SURVEY §8c/S3: the parameter of ehf:189/191 kept as torch.bfloat16): W_bf16
 * holds bf16 bit patterns, w_batch_stride counts bf16 elements, everything else as above.  A, Y
 * and the accumulation stay fp32.  A bf16 value is its own high plane, so the split kernel
 * (K a multiple of 4 in [16,128]) needs three plane products per term instead of six; the result
 * is bit-identical to tmgcn_gemm_f32 on the fp32-widened weight.
 */
int tmgcn_gemm_bf16w_f32(const float* A, const uint16_t* W_bf16, float* Y, float* pre_act,
                         int64_t R, int32_t K, int32_t Nf, int32_t trans_w,
                         int64_t rows_per_batch, int64_t w_batch_stride, int32_t act, int32_t algo,
                         void* stream);

/* Backward of P3 with respect to the weight (autograd of ehf:222 etc.):
 *   dW_b[k][n] = sum_{r in batch b} A[r][k] * dY[r][n]
 * workspace: tmgcn_gemm_dw_workspace_bytes() bytes of scratch on the device
 * (partial slabs, reduced in a fixed order: bitwise reproducible).
 * For K, Nf >= 16 (multiples of 4, 16-byte aligned operands) the products run as v_mfma_f32_32x32x16_bf16
 * on three bf16 planes per operand, x = hi + mid + lo exactly, keeping the six plane products above
 * 2^-24 |a·b|: the accuracy of an fp32 FMA chain at 2.7x the f32 MFMA rate (algo = TMGCN_DW_AUTO;
 * TMGCN_DW_F32MFMA selects the exact-f32 MFMA kernel for this call).
 */
int64_t tmgcn_gemm_dw_workspace_bytes(int64_t R, int32_t K, int32_t Nf, int64_t rows_per_batch);
int tmgcn_gemm_dw_f32(const float* A, const float* dY, float* dW,
                      int64_t R, int32_t K, int32_t Nf, int64_t rows_per_batch, int32_t algo,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* The same with the activation gradient folded in (ABI 4):  dW_b = A_bᵀ · (dY ⊙ act'(pre_act))  — autograd of
 * act(A·W) (ehf:330-334, 486) with respect to W without the [T,N,F] pass of tmgcn_act_bwd_f32 in between.  Narrow
 * layers only (tmgcn_gemm_dw_act_supported: even K, Nf <= 8 — the reference's 2x6 / 6x6); same workspace. */
int tmgcn_gemm_dw_act_supported(int32_t K, int32_t Nf);
int tmgcn_gemm_dw_act_f32(const float* A, const float* dY, const float* pre_act, int32_t act, float* dW,
                          int64_t R, int32_t K, int32_t Nf, int64_t rows_per_batch,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* ---- layers 1 + 2 of the narrow 2-layer models, fused (ABI 4) --------------------------------------
 *   forward   Z = act2( (Â ⋆ act1(H·W1)) · W2 )     ehf:330-335 + 348-349 (EmbeddingGCN2, default branch), ehf:486-487 (KWGCN)
 *   backward  dW1 = Σ_r H[r]ᵀ · ( ((Âᵀ ⋆ dZ')·W2ᵀ)[r] ⊙ act1'(H[r]·W1) ),   dZ' = dZ ⊙ act2'(pre2)
 * H [R][2] is the model's cached constant (AtXt / AX, ehf:293 / 464): the [R][F] intermediates (layer-1 output, its
 * pre-activation, dY, dP) are never stored.  Shared weights W1 [2][F], W2 [F][Nf]; 2 -> even F <= 8 -> even Nf <= 8
 * (tmgcn_layer12_supported).  Against tmgcn_gemm_f32 + tmgcn_spmm_gemm_f32: another fp32 summation order per row (<= 1e-6).
 * AX (optional) receives Â ⋆ act1(H·W1) for dW2 = AXᵀ·dZ' (tmgcn_gemm_dw_f32); pre2 the pre-activation of layer 2
 * when act2 is not none.  The backward takes the TRANSPOSED batched CSR.
 * row_blocks / n_row_blocks (optional, both 0 = blocks of 256 consecutive rows, ascending): a partition of the rows for
 *   the entry-major kernels — n_row_blocks pairs (first row, number of rows <= 256) of int64 that together cover every row
 *   exactly once (N >= 256: a block then holds at most one slice boundary).  A caller that knows its row lengths
 *   (csr.BatchedCSR.row_blocks) cuts the blocks so that none holds more than about one 1 024-entry tile: with real, skewed
 *   data the longest block otherwise sets the launch time.  The FORWARD starts the blocks in list order: the heaviest first
 *   (heavy blocks started last leave the chip idle behind them).  The BACKWARD's resident thread blocks draw their row blocks
 *   from min(TMGCN_L12_RUNS, n_row_blocks) RUNS of the list — run g = entries [n·g / runs, n·(g + 1) / runs) — each run worked on
 *   by the thread blocks of one XCD, first entry to last: its caller fills each run with NEIGHBOURING row blocks (an XCD's L2
 *   then sees whole slices and fetches the rows they gather once instead of every XCD fetching them) and lists the heaviest
 *   first inside a run (csr.BatchedCSR.row_block_runs).  Any list that covers
 *   the rows is correct; the order is performance only.  The backward takes its
 *   entry-major kernel whenever a partition is given (otherwise only for fewer than 4 entries per row): a caller passes one for
 *   skewed adjacencies — hub rows — and none for evenly filled ones, where lanes-per-row is the faster walk above 4 entries per
 *   row.  The forward's and the backward's partitions are independent (the backward's is over the TRANSPOSED rows); results
 *   do not depend on the partition in the forward (whole rows) and are bit-reproducible for a given partition in the
 *   backward (dW1 / dW2 are summed per row block, the row blocks' sums in a fixed order — whichever thread block worked on which).
 * tmgcn_layer12_bwd_workspace_bytes: for the call's widths, n_rows and n_row_blocks (0 without a partition).
 * AX / dW2 of the backward (optional, both or neither): see tmgcn_layer12_bwd_forms_dw2 below. */
#define TMGCN_L12_RUNS 16
int tmgcn_layer12_supported(int32_t K0, int32_t F, int32_t Nf);
int tmgcn_layer12_fwd_f32(const int64_t* rowptr, const int32_t* col, const float* val, const float* H,
                          const float* W1, int32_t act1, const float* W2, int32_t act2, int64_t n_rows,
                          int32_t N, int32_t K0, int32_t F, int32_t Nf, float* Z, float* AX, float* pre2,
                          float avg_nnz_per_row, const int64_t* row_blocks, int32_t n_row_blocks, void* stream);
/* 1 when the fused forward is the faster route for the shape (slices of >= 256 nodes: the entry-major kernel; small dense
 * slices whose layer-1 output is formed once per node in LDS; short rows); 0: form act1(H·W1) with tmgcn_gemm_f32 and call
 * tmgcn_spmm_gemm_f32 (same Z up to fp32 summation order). */
int tmgcn_layer12_fwd_pays(int64_t n_rows, int32_t N, int32_t F, float avg_nnz_per_row);
int64_t tmgcn_layer12_bwd_workspace_bytes(int32_t K0, int32_t F, int32_t Nf, int64_t n_rows, int32_t n_row_blocks);
/* 1 when tmgcn_layer12_bwd_f32 will take its entry-major kernel for this call (slices of >= 256 nodes that are sparse —
 * fewer than 4 entries per row — or come with a partition): that kernel can form dW2 = AXᵀ·(dZ ⊙ act2'(pre2)) — layer 2's
 * weight gradient, ehf:348-349's autograd — in the same launch, from the AX the forward stored: pass AX and dW2 then, and
 * leave tmgcn_gemm_dw(_act)_f32 out.  0: AX and dW2 must be NULL. */
int tmgcn_layer12_bwd_forms_dw2(int64_t n_rows, int32_t N, int32_t F, int32_t Nf, float avg_nnz_per_row, int32_t n_row_blocks);
int tmgcn_layer12_bwd_f32(const int64_t* t_rowptr, const int32_t* t_col, const float* t_val, const float* dZ,
                          const float* pre2, const float* H, const float* W1, int32_t act1, const float* W2,
                          int32_t act2, int64_t n_rows, int32_t N, int32_t K0, int32_t F, int32_t Nf,
                          float* dW1, const float* AX, float* dW2, float avg_nnz_per_row, const int64_t* row_blocks,
                          int32_t n_row_blocks, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- P5: pointwise non-linearity between layers (ehf:284-289, 332-334, 486) -------
 *   fwd: y = act(x);   bwd: dx = dy * act'(x)   (x = the pre-activation input)
 */
int tmgcn_act_fwd_f32(const float* x, float* y, int64_t n, int32_t act, void* stream);
int tmgcn_act_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, int32_t act,
                      void* stream);

/* ---- P4: edge head  (ehf:228-232, 351-355, 491-495) --------------------------------
 *   out[e][c] = sum_{f<F} Z[src[e]][f] * U[f][c] + sum_{f<F} Z[dst[e]][f] * U[F+f][c]
 * src/dst are the flat row indices t*N+node built at ehf:196-198.  Z is [R][F], U is [2F][C].
 * Supported when tmgcn_edge_head_supported(F, C) != 0 (F <= 256, C <= 8); wider heads use the
 * caller's own gather + GEMM.
 * Backward (autograd of the same statements), no atomics, fixed summation order:
 *   dZ[r][f] (every row written, rows without edges get 0) and dU[2F][C].
 *   eptr[R+1] / eidx[2E]: inverted edge index — the entries of row r are eidx[eptr[r]..eptr[r+1]),
 *   entry = 2*edge + role (role 0: r is the edge's src, 1: its dst); built once per edge set.
 *   dZ or dU may be NULL to skip that gradient.
 */
int tmgcn_edge_head_supported(int32_t F, int32_t C);
int tmgcn_edge_head_fwd_f32(const float* Z, const int64_t* src, const int64_t* dst,
                            const float* U, float* out, int64_t E, int32_t F, int32_t C,
                            void* stream);
int64_t tmgcn_edge_head_bwd_workspace_bytes(int64_t E, int32_t F, int32_t C);
int tmgcn_edge_head_bwd_f32(const float* Z, const int64_t* src, const int64_t* dst,
                            const float* U, const float* dout,
                            const int64_t* eptr, const int64_t* eidx,
                            float* dZ, float* dU, int64_t R, int64_t E, int32_t F, int32_t C,
                            void* workspace, int64_t workspace_bytes, void* stream);
/* The same two entry points with 32-bit index arrays (src, dst, eptr, eidx): for edge sets whose
 * rows and 2·E fit 31 bits — every configuration of the reference's experiments — the index
 * arrays are half the bytes these gather-bound kernels move.  Same results, bit for bit. */
int tmgcn_edge_head_fwd_i32_f32(const float* Z, const int32_t* src, const int32_t* dst,
                                const float* U, float* out, int64_t E, int32_t F, int32_t C,
                                void* stream);
int tmgcn_edge_head_bwd_i32_f32(const float* Z, const int32_t* src, const int32_t* dst,
                                const float* U, const float* dout,
                                const int32_t* eptr, const int32_t* eidx,
                                float* dZ, float* dU, int64_t R, int64_t E, int32_t F, int32_t C,
                                void* workspace, int64_t workspace_bytes, void* stream);

/* ---- class-weighted cross entropy, mean reduction (opt-in) -----------------------------
 * Same value as  nn.CrossEntropyLoss(weight=w)(logits, target)  used by every experiment script
 * (experiment_reddit_our_link_prediction.py:69, 79): loss = Σ w[t]·nll / Σ w[t].  One streaming
 * pass each way with fp64 block sums in fixed order; C <= 8.  stats_out: 2 doubles {Σ w·nll, Σ w}
 * kept by the caller for the backward.  A target equal to ignore_index (outside [0, C); the
 * criterion's default is -100) carries no weight and gets a zero gradient.  Any other target
 * outside [0, C) is corrupt input: the loss and every gradient come out NaN (torch device-asserts
 * there) — loud without a host synchronisation.
 */
int64_t tmgcn_wce_workspace_bytes(int64_t E);
int tmgcn_wce_fwd_f32(const float* logits, const int64_t* target, const float* weight, int64_t E,
                      int32_t C, int64_t ignore_index, float* loss_out, double* stats_out, void* workspace,
                      int64_t workspace_bytes, void* stream);
int tmgcn_wce_bwd_f32(const float* logits, const int64_t* target, const float* weight,
                      const double* stats, const float* grad_loss, int64_t E, int32_t C,
                      int64_t ignore_index, float* dlogits, void* stream);

/* ---- one-pass edge head + weighted cross entropy + all gradients (ABI 4) ----------------
 * loss = nn.CrossEntropyLoss(weight=w)(  [Z[src], Z[dst]] · U ,  target )  and, when dU != NULL, the
 * gradients autograd derives for it (upstream gradient 1), in ONE launch — the per-epoch statements
 *   ehf:228-232 / 351-355 / 491-495 (gather, cat, ·U),
 *   experiment_reddit_our_link_prediction.py:69, 79 (criterion, loss.backward())
 * for the narrow heads of the reference's experiments: even F <= 8, 2 <= C <= 4, 32-bit indices
 * (tmgcn_head_loss_supported).  Row-centric over the inverted edge index; no logits / dlogits arrays,
 * no atomics, the tail reduction done by the last block: bitwise reproducible.
 *   eptr[R+1], ent[2E]   the inverted edge index of tmgcn_edge_head_bwd_i32_f32 (entry = 2*edge + role);
 *                        ent is read only when the logits are stored
 *   other[2E]            row index of the OTHER endpoint of each entry's edge
 *   arow[n_active][4]    (row, first entry, end entry, part) of every row with entries, ascending, 16-byte aligned;
 *                        n_active < 2^31.  part = 0: the whole row, i.e. (row, eptr[row], eptr[row+1], 0).  A caller may
 *                        SPLIT a long row (a hub of the labelled edges: a group of at most 16 lanes walks a row's entries) into
 *                        several consecutive ranges, part = 1, 2, … numbered over ALL split rows of the plan: every sum this
 *                        kernel forms is linear in the entries, so parts are independent — except that a part's share of dZ[row]
 *                        goes to row R + part - 1 of dZ (dZ then has R + n_parts rows) and tmgcn_head_loss_combine_f32 adds the
 *                        parts into dZ[row] afterwards, in order.  (K = 2 stores no dZ: nothing to combine.)
 *   meta[2E]             role << 7 | target class of each entry's edge (0..C-1; 127 = ignored: no weight, no gradient)
 *   class_count[C]       number of labelled edges per class: Σ_e w[t_e] = Σ_c class_count[c]·w[c]
 *   grad_scale           NULL, or one float on the device: the gradients are multiplied by it (the upstream
 *                        gradient of the loss, when the gradient launch runs in the caller's backward)
 *   logits               [E][C] or NULL: the logits as a by-product
 *   loss                 one float, or NULL when only the gradients are wanted
 *   dZ [R][F], dU [2F][C]   or both NULL (loss only).  Rows without a labelled edge get dZ = 0.
 *   K = 2 ("fold", the 1-layer model ehf:222): Z is AtXt [R][2] and W_fold [2][F] the shared weight;
 *       Z = AtXt·W is recomputed per row, dZ is not stored and dW [2][F] = Σ_r AtXt[r]ᵀ·dZ[r] is returned.
 *   workspace            tmgcn_head_loss_workspace_bytes(F, C, K) bytes
 *   sync                 TMGCN_SYNC_INTS int32 (the blocks' two-level hand-off counters), ZERO before the first
 *                        launch; the kernel leaves them zero.  Launches that share a sync block must not overlap.
 *                        NULL = a block of the library's own pool.
 */
#define TMGCN_SYNC_INTS 272
int tmgcn_head_loss_supported(int32_t F, int32_t C, int32_t K);
/* lanes that will share one arow entry (1, 4 or 16: chosen from the mean number of entries per arow entry; a lane takes its
 * entries 2 (one lane) or 8 at a time): a plan that splits rows sizes its parts from this — about eight trips of the group */
int tmgcn_head_loss_lanes(int64_t E, int64_t n_active);
/* srow [n_split][4] = (row, first part - 1, number of parts, 0) of every split row: dZ[row] = Σ_k dZ[R + first + k], k ascending */
int tmgcn_head_loss_combine_f32(const int32_t* srow, int32_t n_split, float* dZ, int64_t R, int32_t F, void* stream);
int64_t tmgcn_head_loss_workspace_bytes(int32_t F, int32_t C, int32_t K);
int tmgcn_head_loss_f32(const float* Z, const float* W_fold, int32_t K, const float* U,
                        const int32_t* eptr, const int32_t* arow, int64_t n_active, const int32_t* ent,
                        const int32_t* other, const uint8_t* meta, const int64_t* class_count,
                        const float* weight, const float* grad_scale, int64_t R, int64_t E, int32_t F,
                        int32_t C, float* logits, float* loss, float* dZ, float* dU, float* dW,
                        void* workspace, int64_t workspace_bytes, int32_t* sync, void* stream);
/* The folded 1-layer model's WHOLE training step in one launch: tmgcn_head_loss_f32 with K = 2 (loss, dU, dW as above),
 * whose last block then applies t.optim.SGD's update (experiment_reddit_our_link_prediction.py:68, 80) to U [2F][C] and
 * W_fold [2][F] in place — the arithmetic of tmgcn_sgd_step, element for element.  buf_U / buf_W: the momentum buffers
 * (NULL when momentum == 0); first_step: the buffers are written, not read (torch's first step).  The kernel's own reads of
 * U and W are complete when the update starts (every other block has handed its slab over). */
typedef struct TmgcnSgd {
  float* buf_U;
  float* buf_W;
  float lr, momentum, dampening, weight_decay;
  int32_t nesterov, maximize, first_step;
} TmgcnSgd;
int tmgcn_head_loss_sgd_f32(const float* Z, float* W_fold, int32_t K, float* U, const int32_t* eptr, const int32_t* arow,
                            int64_t n_active, const int32_t* other, const uint8_t* meta, const int64_t* class_count,
                            const float* weight, int64_t R, int64_t E, int32_t F, int32_t C, float* loss, float* dU,
                            float* dW, const TmgcnSgd* sgd, void* workspace, int64_t workspace_bytes, int32_t* sync,
                            void* stream);

/* ---- optimizer step of all parameters in one launch (ABI 4) ------------------------------------
 * torch.optim.SGD(params, lr, momentum, dampening, weight_decay, nesterov, maximize).step() for up to 16 tensors
 * (experiment_reddit_our_link_prediction.py:68, 80):  g (+ wd·p);  buf = first_step ? g : m·buf + (1-d)·g;
 * p -= lr·(nesterov ? g + m·buf : buf).  HOST arrays of DEVICE pointers; all tensors fp32 (bf16 = 0) or all
 * stored in bf16 (bf16 = 1: widened on load, fp32 arithmetic, rounded to nearest even once).
 */
int tmgcn_sgd_step(void* const* params, const void* const* grads, void* const* momentum_bufs, const int64_t* numel,
                   int32_t n, int32_t bf16, float lr, float momentum, float dampening, float weight_decay,
                   int32_t nesterov, int32_t maximize, int32_t first_step, void* stream);
/* bf16 <-> fp32 of up to 16 tensors in one launch (to_bf16 = 0: src bf16 -> dst fp32; 1: src fp32 -> dst bf16, round to
 * nearest even).  HOST arrays of DEVICE pointers. */
int tmgcn_cast_multi(const void* const* src, void* const* dst, const int64_t* numel, int32_t n, int32_t to_bf16,
                     void* stream);
/* out_a = g·a, out_b = g·b (g: one float on the device — the upstream gradient of a scalar loss) */
int tmgcn_scale2_f32(const float* g, const float* a, float* out_a, int64_t na, const float* b, float* out_b,
                     int64_t nb, void* stream);

/* ---- adjacency pipeline on the device ----------------------------------
 * Replaces the reference's offline preprocessing loops: read_data.py:88-111 (symmetrise),
 * :116-125 (edge-life window), :130-169 (add I, D^-1/2 · D^-1/2), :204-223 (sparse M-product),
 * MATLAB read_data.m:172-209, and the list-of-COO ingest ehf:560-574.
 * A batched COO is (key, val) with ONE 64-bit key per entry:  key = (slice*N + row)*N + col.
 * Each step either fans entries out ("expand": output arrays sized n * fan-out by the caller;
 * unused slots carry the sentinel key ~0 and value 0) or is elementwise; tmgcn_coo_sort_reduce
 * sorts by key and sums equal keys in sorted order (reproducible).  A sentinel run, if any, ends
 * up as the LAST reduced entry (key ~0): the caller drops it.
 */
int tmgcn_adj_make_keys(const int64_t* slice, const int64_t* row, const int64_t* col, int64_t n,
                        int64_t N, uint64_t* key, void* stream);
int64_t tmgcn_coo_sort_reduce_workspace_bytes(int64_t n);
int tmgcn_coo_sort_reduce(const uint64_t* keys_in, const float* vals_in, int64_t n, int32_t key_bits,
                          uint64_t* keys_out, float* vals_out, int64_t* n_out_dev,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* out: 2n entries — (t,i,j,v/2) and (t,j,i,v/2) */
int tmgcn_adj_symmetrise(const uint64_t* key, const float* val, int64_t n, int64_t N,
                         uint64_t* okey, float* oval, void* stream);
/* out: n*window entries — the entry of slice t repeated in slices t .. t+window-1 (< T) */
int tmgcn_adj_edge_life(const uint64_t* key, const float* val, int64_t n, int64_t N, int32_t T,
                        int32_t window, uint64_t* okey, float* oval, void* stream);
/* out: TN entries (t,i,i,1) */
int tmgcn_adj_identity(int64_t TN, int64_t N, uint64_t* okey, float* oval, void* stream);
/* in place on a sorted, reduced COO: val *= d[row]*d[col], d = 1/sqrt(row sum); also returns
 * rowptr[TN+1] and dinv[TN] */
int tmgcn_adj_normalise(const uint64_t* key, float* val, int64_t n, int64_t N, int64_t TN,
                        int64_t* rowptr, float* dinv, void* stream);
/* out: n*(band_lo+band_hi+1) entries — (k,r,c, M[k][j]*v) for the rows k of column j inside the band */
int tmgcn_adj_mproduct_expand(const uint64_t* key, const float* val, int64_t n, int64_t N, int32_t T,
                              const float* M, int32_t ldm, int32_t band_lo, int32_t band_hi,
                              uint64_t* okey, float* oval, void* stream);
/* sorted keys -> rowptr[TN+1] (binary search of r*N) and col[n] (key mod N) */
/* The same product as a SEGMENTED MERGE of the column-sorted CSR rows (ABI 3): output row (k, r) is the
 * W-way merge of rows r of the slices the band of M reaches from k, contributions to one column
 * summed in fp64 in a fixed order and rounded once.  No expansion, no sort: the only memory is the
 * output.  Two passes: _count writes the output row lengths as out_count[r+1] (out_count[0] = 0;
 * the caller's inclusive prefix sum of it IS the output rowptr), _fill writes columns and values.
 * Needs band_lo + band_hi + 1 <= 64 (wider: tmgcn_adj_mproduct_expand + tmgcn_coo_sort_reduce).
 *   read_data.py:204-223 func_MProduct, SBM_our.py:78-86, read_data.m:207-209 */
int tmgcn_adj_mproduct_merge_count(const int64_t* rowptr, const int32_t* col, int64_t TN, int32_t N, int32_t T,
                                   const float* M, int32_t ldm, int32_t band_lo, int32_t band_hi,
                                   int64_t* out_count, void* stream);
int tmgcn_adj_mproduct_merge_fill(const int64_t* rowptr, const int32_t* col, const float* val, int64_t TN,
                                  int32_t N, int32_t T, const float* M, int32_t ldm, int32_t band_lo,
                                  int32_t band_hi, const int64_t* out_rowptr, int32_t* out_col, float* out_val,
                                  void* stream);
int tmgcn_adj_keys_to_csr(const uint64_t* key, int64_t n, int64_t N, int64_t TN, int64_t* rowptr,
                          int32_t* col, void* stream);
/* keys (slice, col, row) of every entry of a batched CSR: sort them to get the per-slice transpose */
int tmgcn_adj_transpose_keys(const int64_t* rowptr, const int32_t* col, int64_t TN, int64_t N,
                             uint64_t* okey, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TMGCN_H */

// torch_ops.cpp — the PyTorch-ROCm extension layer over the C-ABI (include/tmgcn.h):
// TORCH_LIBRARY(tmgcn) operators with registered autograd, so that the Python host
// (tm-gcn_amd/ops.py) reaches the gfx950 kernels through torch.ops.tmgcn.* — one dispatcher
// call per operator instead of a ctypes argument list per launch, and the backward formulas of
// the reference's autograd as C++ autograd functions.
//
//   kernel-level ops (one C-ABI launch each, no autograd)
//     tmgcn::mtransform          tmgcn_mtransform_f32              ehf:204, 308, 346, 404; Minv ehf:224
//     tmgcn::spmm_csr_batched    tmgcn_spmm_csr_batched_f32_hint   ehf:206-207, 303-304, 310-311, 471-472
//     tmgcn::spmm_gemm(_out)     tmgcn_spmm_gemm_f32               the two statements above + ehf:222 in one launch
//     tmgcn::bgemm               tmgcn_gemm_f32                    ehf:222, 330, 344, 349, 486-489
//     tmgcn::bgemm_dW            tmgcn_gemm_dw_f32                 autograd of ehf:222
//     tmgcn::edge_head_fwd/bwd   tmgcn_edge_head_*_f32             ehf:228-232, 351-355, 491-495
//     tmgcn::act_fwd/bwd         tmgcn_act_*_f32                   ehf:284-289
//     tmgcn::wce_fwd/bwd         tmgcn_wce_*_f32                   experiment_reddit_our_link_prediction.py:69, 79
//   differentiable ops (registered under the Autograd key)
//     tmgcn::m_transform, tmgcn::spmm, tmgcn::feature_gemm, tmgcn::spmm_feature_gemm,
//     tmgcn::edge_head, tmgcn::activation, tmgcn::weighted_ce
//
// No kernels live here: every launch goes through the C-ABI shared library (libtmgcn_hip.so),
// on torch's current HIP stream.  Errors surface as RuntimeError (TORCH_CHECK), the reference's
// convention.  Built by `make -C tm-gcn_amd/csrc torch` (g++; __graft_entry__.build()).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <mutex>
#include <tuple>
#include <vector>

#include "tmgcn.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using OptTensor = c10::optional<Tensor>;

inline void* ptr(const Tensor& t) {
  return (t.defined() && t.numel()) ? const_cast<void*>(t.const_data_ptr()) : nullptr;
}
inline void* stream_of(const Tensor& t) {
  return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}
inline void want(const Tensor& t, const char* name, at::ScalarType dt = at::kFloat) {
  TORCH_CHECK(t.defined(), name, ": undefined tensor");
  TORCH_CHECK(t.is_cuda(), name, ": expected a ROCm (cuda) tensor, got device ", t.device(),
              "; the TM-GCN layer has no CPU path");
  TORCH_CHECK(t.scalar_type() == dt, name, ": expected dtype ", dt, ", got ", t.scalar_type());
  TORCH_CHECK(t.is_contiguous(), name, ": expected a contiguous tensor");
}
inline void ok(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed (status ", rc, "): ", tmgcn_last_error());
}
inline Tensor none_like(const Tensor& t) { return at::empty({0}, t.options()); }
// a contiguous view can start anywhere inside its allocation; the few entry points that need
// 16-byte aligned rows get a fresh (allocator-aligned) copy in that rare case
inline Tensor aligned16(const Tensor& t) {
  return (!t.defined() || reinterpret_cast<uintptr_t>(t.const_data_ptr()) % 16 == 0) ? t : t.clone();
}
inline bool has(const Tensor& t) { return t.defined() && t.numel() > 0; }

// ---------------------------------------------------------------------------------------
// kernel-level operators
// ---------------------------------------------------------------------------------------
Tensor mtransform(const Tensor& M, const Tensor& X, bool transpose, int64_t row_off, int64_t col_off,
                  int64_t T_out, int64_t band_lo, int64_t band_hi, int64_t x_group_rows,
                  int64_t y_group_rows) {
  want(M, "mtransform M");
  want(X, "mtransform X");
  TORCH_CHECK(M.dim() == 2 && M.size(0) == M.size(1), "mtransform: M must be square");
  TORCH_CHECK(X.dim() >= 1, "mtransform: X needs a leading time mode");
  c10::DeviceGuard g(X.device());
  const int64_t T_in = X.size(0);
  if (T_out < 0) T_out = T_in;
  const int64_t C = T_in ? X.numel() / T_in : 0;
  auto sizes = X.sizes().vec();
  sizes[0] = T_out;
  Tensor Y = at::empty(sizes, X.options());
  ok(tmgcn_mtransform_f32((const float*)ptr(M), (int32_t)M.size(0), (int32_t)M.size(0), transpose ? 1 : 0,
                          (int32_t)row_off, (int32_t)col_off, (int32_t)T_out, (int32_t)T_in,
                          (int32_t)band_lo, (int32_t)band_hi, (const float*)ptr(X), (float*)ptr(Y), C,
                          (int32_t)x_group_rows, (int32_t)y_group_rows, stream_of(X)),
     "tmgcn_mtransform_f32");
  return Y;
}

// Column-window form: X [T_in, n, F] and Y [T_out, n, F] may be views whose slices are further apart
// than n*F (e.g. Xt[:, c0:c1, :] of a resident [T, N, F] tensor); inside a slice they are dense.
// The consumer of the node-chunked all-gather (dist.py) — a chunk is transformed into / out of its
// column window of the resident tensor, the replicated [T, N, F] tensor never exists.
void mtransform_out(const Tensor& M, const Tensor& X, Tensor Y, bool transpose, int64_t row_off, int64_t col_off,
                    int64_t band_lo, int64_t band_hi, int64_t x_group_rows, int64_t y_group_rows) {
  want(M, "mtransform M");
  auto window = [](const Tensor& t, const char* name) {
    TORCH_CHECK(t.defined() && t.is_cuda() && t.scalar_type() == at::kFloat, name, ": expected an fp32 ROCm tensor");
    TORCH_CHECK(t.dim() == 3, name, ": expected [T, n, F]");
    TORCH_CHECK(t.size(0) == 0 || t.numel() == 0 ||
                    (t.stride(2) == 1 && t.stride(1) == t.size(2) && (t.size(0) == 1 || t.stride(0) >= t.size(1) * t.size(2))),
                name, ": slices must be dense [n, F] blocks (strides ", t.strides(), ")");
  };
  window(X, "mtransform_out X");
  window(Y, "mtransform_out Y");
  TORCH_CHECK(M.dim() == 2 && M.size(0) == M.size(1), "mtransform: M must be square");
  TORCH_CHECK(X.size(1) == Y.size(1) && X.size(2) == Y.size(2), "mtransform_out: X ", X.sizes(), " and Y ", Y.sizes(),
              " differ in their column extent");
  c10::DeviceGuard g(X.device());
  const int64_t T_in = X.size(0), T_out = Y.size(0), C = X.size(1) * X.size(2);
  const int64_t ldx = T_in > 1 ? X.stride(0) : C, ldy = T_out > 1 ? Y.stride(0) : C;
  ok(tmgcn_mtransform_ld_f32((const float*)ptr(M), (int32_t)M.size(0), (int32_t)M.size(0), transpose ? 1 : 0,
                             (int32_t)row_off, (int32_t)col_off, (int32_t)T_out, (int32_t)T_in, (int32_t)band_lo,
                             (int32_t)band_hi, (const float*)ptr(X), ldx, (float*)ptr(Y), ldy, C, (int32_t)x_group_rows,
                             (int32_t)y_group_rows, stream_of(X)),
     "tmgcn_mtransform_ld_f32");
}

void check_csr(const Tensor& rowptr, const Tensor& col, const Tensor& val, const Tensor& X, int64_t N,
               const char* who) {
  want(rowptr, "rowptr", at::kLong);
  want(col, "col", at::kInt);
  want(val, "val");
  TORCH_CHECK(X.dim() == 3 && X.size(1) == N && (X.size(0) * N + 1) == rowptr.numel(), who, ": X ",
              X.sizes(), " does not match adjacency T=", N ? (rowptr.numel() - 1) / N : 0, " N=", N);
  TORCH_CHECK(val.device() == X.device(), who, ": adjacency and X live on different devices");
}

// A CSR's giant-row plan (include/tmgcn.h "Giant rows"; csr.BatchedCSR.giant_plan): the rows (int64 [n]) and the chunk
// arrays (int32 [n + 1 + m]) on the device, plus the workspace this call allocates for the partial sums.
struct Giant {
  const int64_t* rows = nullptr;
  const int32_t* chunks = nullptr;
  int32_t n = 0, m = 0;
  Tensor ws;
  float* ws_ptr() const { return ws.defined() ? (float*)ws.data_ptr() : nullptr; }
  int64_t ws_bytes() const { return ws.defined() ? ws.numel() * 4 : 0; }
};
Giant giant_of(const OptTensor& g_rows, const OptTensor& g_chunks, const Tensor& X, const char* who) {
  Giant g;
  if (!g_rows.has_value() || !g_chunks.has_value() || g_rows->numel() == 0) return g;
  TORCH_CHECK(g_rows->is_cuda() && g_chunks->is_cuda() && g_rows->scalar_type() == at::kLong && g_chunks->scalar_type() == at::kInt &&
                  g_rows->is_contiguous() && g_chunks->is_contiguous() && g_chunks->numel() > 2 * g_rows->numel(),
              who, ": a giant-row plan is (int64 rows [n], int32 chunks [n + 1 + m]) on the device");
  g.rows = (const int64_t*)g_rows->const_data_ptr();
  g.chunks = (const int32_t*)g_chunks->const_data_ptr();
  g.n = (int32_t)g_rows->numel();
  g.m = (int32_t)(g_chunks->numel() - g_rows->numel() - 1);
  g.ws = at::empty({(int64_t)g.m, X.size(2)}, X.options());
  return g;
}

Tensor spmm_csr_batched(const Tensor& rowptr, const Tensor& col, const Tensor& val, const Tensor& X,
                        int64_t N, double avg_nnz_per_row, const OptTensor& giant_rows, const OptTensor& giant_chunks) {
  want(X, "spmm X");
  check_csr(rowptr, col, val, X, N, "spmm");
  c10::DeviceGuard g(X.device());
  Tensor Y = at::empty_like(X);
  const Giant gi = giant_of(giant_rows, giant_chunks, X, "spmm");
  ok(tmgcn_spmm_csr_batched_f32_plan((const int64_t*)ptr(rowptr), (const int32_t*)ptr(col),
                                     (const float*)ptr(val), (const float*)ptr(X), (float*)ptr(Y),
                                     X.size(0) * N, (int32_t)N, (int32_t)X.size(2), (float)avg_nnz_per_row,
                                     gi.rows, gi.chunks, gi.n, gi.m, gi.ws_ptr(), gi.ws_bytes(), stream_of(X)),
     "tmgcn_spmm_csr_batched_f32");
  return Y;
}

struct WShape {
  bool per_slice;
  int64_t wk, wn, stride;
};
WShape w_shape(const Tensor& W, bool trans_w, int64_t T, int64_t K, const char* who) {
  TORCH_CHECK(W.dim() == 2 || W.dim() == 3, who, ": W must be [K,Nf] or [T,K,Nf]");
  WShape s;
  s.per_slice = W.dim() == 3;
  const int64_t a = W.size(-2), b = W.size(-1);
  s.wk = trans_w ? b : a;
  s.wn = trans_w ? a : b;
  s.stride = s.per_slice ? a * b : 0;
  TORCH_CHECK(s.wk == K && (!s.per_slice || W.size(0) == T), who, ": size mismatch, operand [", T, ",*,", K,
              "] W ", W.sizes(), " trans_w=", trans_w);
  return s;
}

void spmm_gemm_launch(const Tensor& rowptr, const Tensor& col, const Tensor& val, const Tensor& X, int64_t N,
                      const Tensor& W, bool trans_w, int64_t act, const Tensor& Y, const Tensor& AX,
                      const Tensor& pre, int64_t grid_reserve, double avg_nnz_per_row, const OptTensor& giant_rows,
                      const OptTensor& giant_chunks) {
  const WShape s = w_shape(W, trans_w, X.size(0), X.size(2), "spmm_gemm");
  const Giant gi = giant_of(giant_rows, giant_chunks, X, "spmm_gemm");
  // average row length (steers the lanes per row of the narrow kernel, hence its summation order):
  // the CALLER's figure for the rows it launches over.  A one-slice view of a larger CSR shares
  // the whole col/val arrays, so col.numel() / n_rows would be T times too large there; a
  // negative value means "unknown" and takes the kernel's default.
  const int64_t n_rows = X.size(0) * N;
  const float avg = (float)avg_nnz_per_row;
  ok(tmgcn_spmm_gemm_f32_plan((const int64_t*)ptr(rowptr), (const int32_t*)ptr(col), (const float*)ptr(val),
                              (const float*)ptr(X), n_rows, (int32_t)N, (int32_t)X.size(2),
                              (const float*)ptr(W), (int32_t)s.wn, trans_w ? 1 : 0, s.per_slice ? N : 0, s.stride,
                              (int32_t)act, (float*)ptr(Y), (float*)ptr(AX), (float*)ptr(pre), (int32_t)grid_reserve,
                              avg, gi.rows, gi.chunks, gi.n, gi.m, gi.ws_ptr(), gi.ws_bytes(), stream_of(X)),
     "tmgcn_spmm_gemm_f32");
}

std::tuple<Tensor, Tensor, Tensor> spmm_gemm(const Tensor& rowptr, const Tensor& col, const Tensor& val,
                                             const Tensor& X, int64_t N, const Tensor& W, bool trans_w,
                                             int64_t act, bool want_ax, bool want_pre, int64_t grid_reserve,
                                             double avg_nnz_per_row, const OptTensor& giant_rows,
                                             const OptTensor& giant_chunks) {
  want(X, "spmm_gemm X");
  want(W, "spmm_gemm W");
  check_csr(rowptr, col, val, X, N, "spmm_gemm");
  c10::DeviceGuard g(X.device());
  const WShape s = w_shape(W, trans_w, X.size(0), X.size(2), "spmm_gemm");
  Tensor Y = at::empty({X.size(0), N, s.wn}, X.options());
  Tensor AX = want_ax ? at::empty(X.sizes(), X.options()) : Tensor();
  Tensor pre = (want_pre && act != TMGCN_ACT_NONE) ? at::empty_like(Y) : Tensor();
  spmm_gemm_launch(rowptr, col, val, aligned16(X), N, W, trans_w, act, Y, AX, pre, grid_reserve, avg_nnz_per_row, giant_rows,
                   giant_chunks);
  return {Y, AX.defined() ? AX : none_like(X), pre.defined() ? pre : none_like(X)};
}

// writes into caller-provided (views of) tensors: the slice-by-slice pipelined multi-GPU path
void spmm_gemm_out(const Tensor& rowptr, const Tensor& col, const Tensor& val, const Tensor& X, int64_t N,
                   const Tensor& W, bool trans_w, int64_t act, Tensor Y, const OptTensor& AX,
                   const OptTensor& pre, int64_t grid_reserve, double avg_nnz_per_row, const OptTensor& giant_rows,
                   const OptTensor& giant_chunks) {
  want(X, "spmm_gemm X");
  want(W, "spmm_gemm W");
  want(Y, "spmm_gemm out Y");
  check_csr(rowptr, col, val, X, N, "spmm_gemm");
  c10::DeviceGuard g(X.device());
  const WShape s = w_shape(W, trans_w, X.size(0), X.size(2), "spmm_gemm");
  TORCH_CHECK(Y.dim() == 3 && Y.size(0) == X.size(0) && Y.size(1) == N && Y.size(2) == s.wn,
              "spmm_gemm: out Y ", Y.sizes(), " != [", X.size(0), ",", N, ",", s.wn, "]");
  Tensor ax = AX.has_value() ? *AX : Tensor(), pr = pre.has_value() ? *pre : Tensor();
  if (ax.defined()) want(ax, "spmm_gemm out AX");
  if (pr.defined()) want(pr, "spmm_gemm out pre");
  TORCH_CHECK(!ax.defined() || reinterpret_cast<uintptr_t>(ax.const_data_ptr()) % 16 == 0,
              "spmm_gemm: the AX output view must start 16-byte aligned");
  spmm_gemm_launch(rowptr, col, val, aligned16(X), N, W, trans_w, act, Y, ax, pr, grid_reserve, avg_nnz_per_row, giant_rows,
                   giant_chunks);
}

std::tuple<Tensor, Tensor> bgemm(const Tensor& A, const Tensor& W, bool trans_w, int64_t act, bool want_pre,
                                 int64_t algo) {
  want(A, "gemm A");
  const bool w_bf16 = W.defined() && W.scalar_type() == at::kBFloat16;  // a parameter stored in bf16
  want(W, "gemm W", w_bf16 ? at::kBFloat16 : at::kFloat);
  TORCH_CHECK(A.dim() == 3, "gemm: A must be [T,N,K]");
  c10::DeviceGuard g(A.device());
  const int64_t T = A.size(0), N = A.size(1), K = A.size(2);
  const WShape s = w_shape(W, trans_w, T, K, "gemm");
  Tensor Y = at::empty({T, N, s.wn}, A.options());
  Tensor pre = (want_pre && act != TMGCN_ACT_NONE) ? at::empty_like(Y) : Tensor();
  if (w_bf16)
    ok(tmgcn_gemm_bf16w_f32((const float*)ptr(A), (const uint16_t*)ptr(W), (float*)ptr(Y), (float*)ptr(pre), T * N,
                            (int32_t)K, (int32_t)s.wn, trans_w ? 1 : 0, s.per_slice ? N : 0, s.stride,
                            (int32_t)act, (int32_t)algo, stream_of(A)),
       "tmgcn_gemm_bf16w_f32");
  else
    ok(tmgcn_gemm_f32((const float*)ptr(A), (const float*)ptr(W), (float*)ptr(Y), (float*)ptr(pre), T * N,
                      (int32_t)K, (int32_t)s.wn, trans_w ? 1 : 0, s.per_slice ? N : 0, s.stride, (int32_t)act,
                      (int32_t)algo, stream_of(A)),
       "tmgcn_gemm_f32");
  return {Y, pre.defined() ? pre : none_like(A)};
}

Tensor bgemm_dW(const Tensor& A, const Tensor& dY, bool per_slice, int64_t algo) {
  want(A, "gemm_dw A");
  want(dY, "gemm_dw dY");
  TORCH_CHECK(A.dim() == 3 && dY.dim() == 3 && A.size(0) == dY.size(0) && A.size(1) == dY.size(1),
              "gemm_dw: A ", A.sizes(), " and dY ", dY.sizes(), " do not match");
  c10::DeviceGuard g(A.device());
  const int64_t T = A.size(0), N = A.size(1), K = A.size(2), Nf = dY.size(2), R = T * N;
  const int64_t rpb = per_slice ? N : 0;
  const int64_t need = tmgcn_gemm_dw_workspace_bytes(R, (int32_t)K, (int32_t)Nf, rpb);
  // scratch from torch's caching allocator: stream-ordered, no synchronisation, graph-capturable
  Tensor ws = at::empty({need > 0 ? need : 1}, A.options().dtype(at::kByte));
  Tensor dW = per_slice ? at::empty({T, K, Nf}, A.options()) : at::empty({K, Nf}, A.options());
  ok(tmgcn_gemm_dw_f32((const float*)ptr(A), (const float*)ptr(dY), (float*)ptr(dW), R, (int32_t)K, (int32_t)Nf,
                       rpb, (int32_t)algo, ptr(ws), ws.numel(), stream_of(A)),
     "tmgcn_gemm_dw_f32");
  return dW;
}

// dW = Aᵀ·(dY ⊙ act'(pre)) in one launch (narrow layers; shared weight or one per slice)
Tensor bgemm_dW_act(const Tensor& A, const Tensor& dY, const Tensor& pre, int64_t act, bool per_slice) {
  want(A, "gemm_dw A");
  want(dY, "gemm_dw dY");
  want(pre, "gemm_dw pre-activation");
  TORCH_CHECK(A.dim() == 3 && dY.dim() == 3 && A.size(0) == dY.size(0) && A.size(1) == dY.size(1) && pre.sizes() == dY.sizes(),
              "gemm_dw_act: A ", A.sizes(), ", dY ", dY.sizes(), " and pre ", pre.sizes(), " do not match");
  c10::DeviceGuard g(A.device());
  const int64_t T = A.size(0), N = A.size(1), K = A.size(2), Nf = dY.size(2), R = T * N;
  const int64_t rpb = per_slice ? N : 0;
  const int64_t need = tmgcn_gemm_dw_workspace_bytes(R, (int32_t)K, (int32_t)Nf, rpb);
  Tensor ws = at::empty({need > 0 ? need : 1}, A.options().dtype(at::kByte));
  Tensor dW = per_slice ? at::empty({T, K, Nf}, A.options()) : at::empty({K, Nf}, A.options());
  ok(tmgcn_gemm_dw_act_f32((const float*)ptr(A), (const float*)ptr(dY), (const float*)ptr(pre), (int32_t)act, (float*)ptr(dW), R,
                           (int32_t)K, (int32_t)Nf, rpb, ptr(ws), ws.numel(), stream_of(A)),
     "tmgcn_gemm_dw_act_f32");
  return dW;
}

Tensor edge_head_fwd(const Tensor& Z2, const Tensor& src, const Tensor& dst, const Tensor& U) {
  want(Z2, "edge_head Z");
  want(U, "edge_head U");
  const bool i32 = src.defined() && src.scalar_type() == at::kInt;  // EdgeIndex keeps 32-bit arrays where they fit
  want(src, "edge_head src", i32 ? at::kInt : at::kLong);
  want(dst, "edge_head dst", i32 ? at::kInt : at::kLong);
  TORCH_CHECK(Z2.dim() == 2 && U.dim() == 2 && U.size(0) == 2 * Z2.size(1), "edge_head: U ", U.sizes(),
              " does not match F=", Z2.size(1));
  TORCH_CHECK(src.numel() == dst.numel(), "edge_head: src and dst differ in length");
  c10::DeviceGuard g(Z2.device());
  const int64_t E = src.numel(), F = Z2.size(1), C = U.size(1);
  Tensor out = at::empty({E, C}, Z2.options());
  if (i32)
    ok(tmgcn_edge_head_fwd_i32_f32((const float*)ptr(Z2), (const int32_t*)ptr(src), (const int32_t*)ptr(dst),
                                   (const float*)ptr(U), (float*)ptr(out), E, (int32_t)F, (int32_t)C, stream_of(Z2)),
       "tmgcn_edge_head_fwd_i32_f32");
  else
    ok(tmgcn_edge_head_fwd_f32((const float*)ptr(Z2), (const int64_t*)ptr(src), (const int64_t*)ptr(dst),
                               (const float*)ptr(U), (float*)ptr(out), E, (int32_t)F, (int32_t)C, stream_of(Z2)),
       "tmgcn_edge_head_fwd_f32");
  return out;
}

std::tuple<Tensor, Tensor> edge_head_bwd(const Tensor& Z2, const Tensor& src, const Tensor& dst, const Tensor& U,
                                         const Tensor& dout, const Tensor& eptr, const Tensor& eidx, bool need_dz,
                                         bool need_du) {
  want(Z2, "edge_head Z");
  want(U, "edge_head U");
  want(dout, "edge_head dout");
  const bool i32 = src.defined() && src.scalar_type() == at::kInt;
  const auto it = i32 ? at::kInt : at::kLong;
  want(src, "edge_head src", it);
  want(dst, "edge_head dst", it);
  want(eptr, "edge_head eptr", it);
  want(eidx, "edge_head eidx", it);
  c10::DeviceGuard g(Z2.device());
  TORCH_CHECK(Z2.dim() == 2 && U.dim() == 2 && U.size(0) == 2 * Z2.size(1), "edge_head_bwd: U ", U.sizes(),
              " does not match F=", Z2.dim() == 2 ? Z2.size(1) : -1);
  TORCH_CHECK(src.numel() == dst.numel(), "edge_head_bwd: src and dst differ in length");
  const int64_t R = Z2.size(0), F = Z2.size(1), C = U.size(1), E = src.numel();
  TORCH_CHECK(dout.dim() == 2 && dout.size(0) == E && dout.size(1) == C, "edge_head_bwd: dout ", dout.sizes(),
              " is not [E=", E, ", C=", C, "]");
  TORCH_CHECK(eptr.numel() == R + 1 && eidx.numel() == 2 * E, "edge_head_bwd: inverted index does not match R=", R,
              " E=", E);
  Tensor dZ = need_dz ? at::empty_like(Z2) : Tensor();
  Tensor dU = need_du ? at::empty_like(U) : Tensor();
  const int64_t need = tmgcn_edge_head_bwd_workspace_bytes(E, (int32_t)F, (int32_t)C);
  Tensor ws = at::empty({need > 0 ? need : 1}, Z2.options().dtype(at::kByte));
  if (i32)
    ok(tmgcn_edge_head_bwd_i32_f32((const float*)ptr(Z2), (const int32_t*)ptr(src), (const int32_t*)ptr(dst),
                                   (const float*)ptr(U), (const float*)ptr(dout), (const int32_t*)ptr(eptr),
                                   (const int32_t*)ptr(eidx), (float*)ptr(dZ), (float*)ptr(dU), R, E, (int32_t)F,
                                   (int32_t)C, ptr(ws), ws.numel(), stream_of(Z2)),
       "tmgcn_edge_head_bwd_i32_f32");
  else
    ok(tmgcn_edge_head_bwd_f32((const float*)ptr(Z2), (const int64_t*)ptr(src), (const int64_t*)ptr(dst),
                               (const float*)ptr(U), (const float*)ptr(dout), (const int64_t*)ptr(eptr),
                               (const int64_t*)ptr(eidx), (float*)ptr(dZ), (float*)ptr(dU), R, E, (int32_t)F,
                               (int32_t)C, ptr(ws), ws.numel(), stream_of(Z2)),
       "tmgcn_edge_head_bwd_f32");
  return {dZ.defined() ? dZ : none_like(Z2), dU.defined() ? dU : none_like(Z2)};
}

Tensor act_fwd(const Tensor& x, int64_t act) {
  want(x, "act x");
  c10::DeviceGuard g(x.device());
  Tensor y = at::empty_like(x);
  ok(tmgcn_act_fwd_f32((const float*)ptr(x), (float*)ptr(y), x.numel(), (int32_t)act, stream_of(x)),
     "tmgcn_act_fwd_f32");
  return y;
}

Tensor act_bwd(const Tensor& x, const Tensor& dy, int64_t act) {
  want(x, "act x");
  want(dy, "act dy");
  TORCH_CHECK(x.numel() == dy.numel(), "act_bwd: size mismatch");
  c10::DeviceGuard g(x.device());
  Tensor dx = at::empty_like(x);
  ok(tmgcn_act_bwd_f32((const float*)ptr(x), (const float*)ptr(dy), (float*)ptr(dx), x.numel(), (int32_t)act,
                       stream_of(x)),
     "tmgcn_act_bwd_f32");
  return dx;
}

std::tuple<Tensor, Tensor> wce_fwd(const Tensor& logits, const Tensor& target, const Tensor& weight,
                                   int64_t ignore_index) {
  want(logits, "wce logits");
  want(weight, "wce weight");
  want(target, "wce target", at::kLong);
  TORCH_CHECK(logits.dim() == 2 && weight.numel() == logits.size(1) && target.numel() == logits.size(0),
              "wce: shapes logits ", logits.sizes(), " target ", target.sizes(), " weight ", weight.sizes());
  c10::DeviceGuard g(logits.device());
  const int64_t E = logits.size(0), C = logits.size(1);
  Tensor loss = at::empty({}, logits.options());
  Tensor stats = at::empty({2}, logits.options().dtype(at::kDouble));
  Tensor ws = at::empty({tmgcn_wce_workspace_bytes(E)}, logits.options().dtype(at::kByte));
  ok(tmgcn_wce_fwd_f32((const float*)ptr(logits), (const int64_t*)ptr(target), (const float*)ptr(weight), E,
                       (int32_t)C, ignore_index, (float*)loss.data_ptr(), (double*)stats.data_ptr(), ptr(ws),
                       ws.numel(), stream_of(logits)),
     "tmgcn_wce_fwd_f32");
  return {loss, stats};
}

Tensor wce_bwd(const Tensor& logits, const Tensor& target, const Tensor& weight, const Tensor& stats,
               const Tensor& g, int64_t ignore_index) {
  want(logits, "wce logits");
  want(g, "wce grad");
  c10::DeviceGuard gd(logits.device());
  Tensor dz = at::empty_like(logits);
  ok(tmgcn_wce_bwd_f32((const float*)ptr(logits), (const int64_t*)ptr(target), (const float*)ptr(weight),
                       (const double*)stats.data_ptr(), (const float*)g.data_ptr(), logits.size(0),
                       (int32_t)logits.size(1), ignore_index, (float*)ptr(dz), stream_of(logits)),
     "tmgcn_wce_bwd_f32");
  return dz;
}

// One-pass edge head + weighted CE (+ gradients, times the device scalar `gscale` when given).  Returns
// (loss [] or empty, logits [E,C] or empty, dZ [R,F] (fold: dW [K,F]) or empty, dU [2F,C] or empty).
std::tuple<Tensor, Tensor, Tensor, Tensor> head_loss_fwd(const Tensor& Z, const OptTensor& W_fold, const Tensor& U,
                                                         const Tensor& eptr, const Tensor& arow, const Tensor& ent,
                                                         const Tensor& other, const Tensor& meta, const Tensor& counts,
                                                         const Tensor& weight, Tensor sync, const OptTensor& gscale,
                                                         bool grad, bool want_logits, bool want_loss, const OptTensor& srow,
                                                         int64_t n_parts) {
  want(Z, "head_loss Z");
  want(U, "head_loss U");
  want(weight, "head_loss weight");
  want(eptr, "head_loss eptr", at::kInt);
  want(arow, "head_loss arow", at::kInt);
  want(ent, "head_loss ent", at::kInt);
  want(other, "head_loss other", at::kInt);
  want(meta, "head_loss meta", at::kByte);
  want(counts, "head_loss class counts", at::kLong);
  want(sync, "head_loss sync", at::kInt);
  const bool fold = W_fold.has_value() && W_fold->defined();
  if (fold) want(*W_fold, "head_loss W");
  const bool scaled = gscale.has_value() && gscale->defined();
  if (scaled) want(*gscale, "head_loss upstream gradient");
  TORCH_CHECK(Z.dim() == 2 && U.dim() == 2, "head_loss: Z must be [R, F] and U [2F, C]");
  const int64_t R = Z.size(0), K = fold ? Z.size(1) : 0, F = fold ? W_fold->size(1) : Z.size(1), C = U.size(1);
  TORCH_CHECK(!fold || (W_fold->dim() == 2 && W_fold->size(0) == K), "head_loss: W ", fold ? W_fold->sizes() : Z.sizes(),
              " does not match AtXt ", Z.sizes());
  TORCH_CHECK(U.size(0) == 2 * F && weight.numel() == C && counts.numel() == C, "head_loss: U ", U.sizes(), " / weight / counts do not match F=",
              F, " C=", C);
  TORCH_CHECK(tmgcn_head_loss_supported((int32_t)F, (int32_t)C, (int32_t)K), "head_loss: unsupported widths F=", F, " C=", C, " K=", K);
  TORCH_CHECK(eptr.numel() == R + 1 && ent.numel() % 2 == 0 && other.numel() == ent.numel() && meta.numel() == ent.numel() &&
                  arow.dim() == 2 && arow.size(1) == 4 && sync.numel() >= TMGCN_SYNC_INTS,
              "head_loss: plan arrays do not match R=", R);
  TORCH_CHECK(grad || want_loss, "head_loss: nothing asked for");
  const int64_t E = ent.numel() / 2;
  c10::DeviceGuard g(Z.device());
  Tensor loss = want_loss ? at::empty({}, Z.options()) : none_like(Z);
  Tensor logits = want_logits ? at::empty({E, C}, Z.options()) : none_like(Z);
  // rows the plan split (hubs of the labelled edges): their parts' shares of dZ go to n_parts scratch rows behind the R real
  // ones and are added up by tmgcn_head_loss_combine_f32 right after the launch (include/tmgcn.h)
  const bool split = srow.has_value() && srow->defined() && srow->numel() > 0 && n_parts > 0;
  if (split) {
    want(*srow, "head_loss srow", at::kInt);
    TORCH_CHECK(srow->dim() == 2 && srow->size(1) == 4, "head_loss: srow must be [n_split, 4]");
  }
  Tensor dZfull = (grad && !fold) ? at::empty({R + (split ? n_parts : 0), F}, Z.options()) : Tensor();
  Tensor dZ = grad ? (fold ? at::empty({K, F}, Z.options()) : dZfull.narrow(0, 0, R)) : none_like(Z);
  Tensor dU = grad ? at::empty_like(U) : none_like(Z);
  const int64_t need = tmgcn_head_loss_workspace_bytes((int32_t)F, (int32_t)C, (int32_t)K);
  Tensor ws = at::empty({need}, Z.options().dtype(at::kByte));
  ok(tmgcn_head_loss_f32((const float*)ptr(Z), fold ? (const float*)ptr(*W_fold) : nullptr, (int32_t)K, (const float*)ptr(U),
                         (const int32_t*)ptr(eptr), (const int32_t*)ptr(arow), arow.size(0), (const int32_t*)ptr(ent),
                         (const int32_t*)ptr(other), (const uint8_t*)ptr(meta), (const int64_t*)ptr(counts),
                         (const float*)ptr(weight), scaled ? (const float*)gscale->data_ptr() : nullptr, R, E, (int32_t)F,
                         (int32_t)C, (float*)ptr(logits), want_loss ? (float*)loss.data_ptr() : nullptr,
                         (grad && !fold) ? (float*)ptr(dZ) : nullptr, grad ? (float*)ptr(dU) : nullptr,
                         (grad && fold) ? (float*)ptr(dZ) : nullptr, ptr(ws), ws.numel(), (int32_t*)sync.data_ptr(), stream_of(Z)),
     "tmgcn_head_loss_f32");
  if (split && grad && !fold)
    ok(tmgcn_head_loss_combine_f32((const int32_t*)ptr(*srow), (int32_t)srow->size(0), (float*)ptr(dZfull), R, (int32_t)F, stream_of(Z)),
       "tmgcn_head_loss_combine_f32");
  return {loss, logits, dZ, dU};
}

// The folded 1-layer model's whole training step in one launch (tmgcn_head_loss_sgd_f32): loss, dW, dU, and the SGD update
// of W and U in place.  No autograd: the caller is an optimizer-aware step (graphs.GraphedTrainStep(fold_optimizer=True)).
std::tuple<Tensor, Tensor, Tensor> head_loss_sgd(const Tensor& Z, Tensor W_fold, Tensor U, const Tensor& eptr, const Tensor& arow,
                                                 const Tensor& other, const Tensor& meta, const Tensor& counts, const Tensor& weight,
                                                 Tensor sync, const OptTensor& buf_W, const OptTensor& buf_U, double lr, double momentum,
                                                 double dampening, double weight_decay, bool nesterov, bool maximize, bool first_step) {
  want(Z, "head_loss_sgd AtXt");
  want(W_fold, "head_loss_sgd W");
  want(U, "head_loss_sgd U");
  want(weight, "head_loss_sgd weight");
  want(eptr, "head_loss_sgd eptr", at::kInt);
  want(arow, "head_loss_sgd arow", at::kInt);
  want(other, "head_loss_sgd other", at::kInt);
  want(meta, "head_loss_sgd meta", at::kByte);
  want(counts, "head_loss_sgd class counts", at::kLong);
  want(sync, "head_loss_sgd sync", at::kInt);
  TORCH_CHECK(Z.dim() == 2 && U.dim() == 2 && W_fold.dim() == 2 && W_fold.size(0) == Z.size(1), "head_loss_sgd: AtXt [R, 2], W [2, F], U [2F, C]");
  const int64_t R = Z.size(0), K = Z.size(1), F = W_fold.size(1), C = U.size(1), E = other.numel() / 2;
  TORCH_CHECK(K == 2 && U.size(0) == 2 * F && weight.numel() == C && counts.numel() == C &&
                  tmgcn_head_loss_supported((int32_t)F, (int32_t)C, (int32_t)K),
              "head_loss_sgd: unsupported widths F=", F, " C=", C, " K=", K);
  TORCH_CHECK(eptr.numel() == R + 1 && meta.numel() == other.numel() && arow.dim() == 2 && arow.size(1) == 4 && sync.numel() >= TMGCN_SYNC_INTS,
              "head_loss_sgd: plan arrays do not match R=", R);
  const bool mom = momentum != 0.0;
  TORCH_CHECK(!mom || (buf_W.has_value() && buf_W->defined() && buf_U.has_value() && buf_U->defined()), "head_loss_sgd: momentum needs both buffers");
  if (mom) {
    want(*buf_W, "head_loss_sgd momentum buffer of W");
    want(*buf_U, "head_loss_sgd momentum buffer of U");
    TORCH_CHECK(buf_W->numel() == W_fold.numel() && buf_U->numel() == U.numel(), "head_loss_sgd: momentum buffers do not match the parameters");
  }
  c10::DeviceGuard g(Z.device());
  Tensor loss = at::empty({}, Z.options()), dW = at::empty_like(W_fold), dU = at::empty_like(U);
  const int64_t need = tmgcn_head_loss_workspace_bytes((int32_t)F, (int32_t)C, (int32_t)K);
  Tensor ws = at::empty({need}, Z.options().dtype(at::kByte));
  TmgcnSgd sgd{mom ? (float*)buf_U->data_ptr() : nullptr, mom ? (float*)buf_W->data_ptr() : nullptr, (float)lr, (float)momentum,
               (float)dampening, (float)weight_decay, nesterov ? 1 : 0, maximize ? 1 : 0, first_step ? 1 : 0};
  ok(tmgcn_head_loss_sgd_f32((const float*)ptr(Z), (float*)W_fold.data_ptr(), (int32_t)K, (float*)U.data_ptr(), (const int32_t*)ptr(eptr),
                             (const int32_t*)ptr(arow), arow.size(0), (const int32_t*)ptr(other), (const uint8_t*)ptr(meta),
                             (const int64_t*)ptr(counts), (const float*)ptr(weight), R, E, (int32_t)F, (int32_t)C, (float*)loss.data_ptr(),
                             (float*)dU.data_ptr(), (float*)dW.data_ptr(), &sgd, ptr(ws), ws.numel(), (int32_t*)sync.data_ptr(), stream_of(Z)),
     "tmgcn_head_loss_sgd_f32");
  return {loss, dW, dU};
}

std::tuple<Tensor, Tensor> scale2(const Tensor& g, const Tensor& a, const Tensor& b) {
  want(g, "scale2 g");
  want(a, "scale2 a");
  want(b, "scale2 b");
  c10::DeviceGuard gd(a.device());
  Tensor oa = at::empty_like(a), ob = at::empty_like(b);
  ok(tmgcn_scale2_f32((const float*)g.data_ptr(), (const float*)ptr(a), (float*)ptr(oa), a.numel(), (const float*)ptr(b),
                      (float*)ptr(ob), b.numel(), stream_of(a)),
     "tmgcn_scale2_f32");
  return {oa, ob};
}

// Optimizer step of up to 16 parameters in one launch (torch.optim.SGD semantics; tmgcn_amd.optim.FusedSGD).
void sgd_step(at::TensorList params, at::TensorList grads, at::TensorList bufs, double lr, double momentum,
              double dampening, double weight_decay, bool nesterov, bool maximize, bool first_step) {
  const int64_t n = (int64_t)params.size();
  TORCH_CHECK(n >= 1 && n <= 16 && (int64_t)grads.size() == n && (momentum == 0.0 || (int64_t)bufs.size() == n),
              "sgd_step: 1..16 parameters with as many gradients (and momentum buffers)");
  const auto dt = params[0].scalar_type();
  TORCH_CHECK(dt == at::kFloat || dt == at::kBFloat16, "sgd_step: fp32 or bf16 parameters");
  void* pp[16];
  const void* gp[16];
  void* bp[16];
  int64_t ne[16];
  for (int64_t k = 0; k < n; ++k) {
    want(params[k], "sgd_step parameter", dt);
    want(grads[k], "sgd_step gradient", dt);
    TORCH_CHECK(grads[k].numel() == params[k].numel() && params[k].device() == params[0].device(), "sgd_step: tensor ", k,
                " does not match");
    pp[k] = params[k].data_ptr();
    gp[k] = grads[k].const_data_ptr();
    bp[k] = nullptr;
    if (momentum != 0.0) {
      want(bufs[k], "sgd_step momentum buffer", dt);
      TORCH_CHECK(bufs[k].numel() == params[k].numel(), "sgd_step: momentum buffer ", k, " does not match");
      bp[k] = bufs[k].data_ptr();
    }
    ne[k] = params[k].numel();
  }
  c10::DeviceGuard g(params[0].device());
  ok(tmgcn_sgd_step(pp, gp, bp, ne, (int32_t)n, dt == at::kBFloat16 ? 1 : 0, (float)lr, (float)momentum, (float)dampening,
                    (float)weight_decay, nesterov ? 1 : 0, maximize ? 1 : 0, first_step ? 1 : 0, stream_of(params[0])),
     "tmgcn_sgd_step");
}

// bf16 parameters -> fp32 copies (one launch for all of them); backward: the fp32 gradients rounded to bf16 (one launch)
std::vector<Tensor> cast_multi(at::TensorList src, bool to_bf16) {
  const int64_t n = (int64_t)src.size();
  TORCH_CHECK(n >= 1 && n <= 16, "cast_multi: 1..16 tensors");
  const void* sp[16];
  void* dp[16];
  int64_t ne[16];
  std::vector<Tensor> out;
  c10::DeviceGuard g(src[0].device());
  for (int64_t k = 0; k < n; ++k) {
    want(src[k], "cast_multi source", to_bf16 ? at::kFloat : at::kBFloat16);
    out.push_back(at::empty_like(src[k], src[k].options().dtype(to_bf16 ? at::kBFloat16 : at::kFloat)));
    sp[k] = src[k].const_data_ptr();
    dp[k] = out[k].data_ptr();
    ne[k] = src[k].numel();
  }
  ok(tmgcn_cast_multi(sp, dp, ne, (int32_t)n, to_bf16 ? 1 : 0, stream_of(src[0])), "tmgcn_cast_multi");
  return out;
}

struct WidenFn : public torch::autograd::Function<WidenFn> {
  static variable_list forward(AutogradContext* ctx, at::TensorList params) {   // TensorList: each element is an input of the node
    at::AutoDispatchBelowADInplaceOrView guard;
    return cast_multi(params, false);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    // a parameter that took no part in the step has no gradient: cast the ones that have
    std::vector<Tensor> have;
    std::vector<size_t> at_;
    for (size_t k = 0; k < grads.size(); ++k)
      if (grads[k].defined()) {
        have.push_back(grads[k].contiguous());
        at_.push_back(k);
      }
    variable_list out(grads.size());
    if (!have.empty()) {
      auto r = cast_multi(have, true);
      for (size_t i = 0; i < at_.size(); ++i) out[at_[i]] = r[i];
    }
    return out;
  }
};
std::vector<Tensor> widen_params_ad(at::TensorList params) { return WidenFn::apply(params); }

bool spmm_gemm_supported(int64_t K, int64_t Nf) { return tmgcn_spmm_gemm_supported((int32_t)K, (int32_t)Nf) != 0; }
bool layer12_supported(int64_t K0, int64_t F, int64_t Nf) { return tmgcn_layer12_supported((int32_t)K0, (int32_t)F, (int32_t)Nf) != 0; }
bool edge_head_supported(int64_t F, int64_t C) { return tmgcn_edge_head_supported((int32_t)F, (int32_t)C) != 0; }
bool head_loss_supported(int64_t F, int64_t C, int64_t K) { return tmgcn_head_loss_supported((int32_t)F, (int32_t)C, (int32_t)K) != 0; }
int64_t abi_version() { return tmgcn_abi_version(); }

// ---------------------------------------------------------------------------------------
// differentiable operators (explicit adjoints: what autograd derives for the reference)
// ---------------------------------------------------------------------------------------
struct MTransformFn : public torch::autograd::Function<MTransformFn> {
  // Y[k] = Σ_j M[ro+k][co+j] X[j]   =>   dX[j] = Σ_k Mᵀ[co+j][ro+k] dY[k]
  static Tensor forward(AutogradContext* ctx, const Tensor& X, const Tensor& M, int64_t band_lo, int64_t band_hi,
                        int64_t row_off, int64_t col_off, int64_t T_out, int64_t xg, int64_t yg) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({M});
    ctx->saved_data["lo"] = band_lo;
    ctx->saved_data["hi"] = band_hi;
    ctx->saved_data["ro"] = row_off;
    ctx->saved_data["co"] = col_off;
    ctx->saved_data["T_in"] = X.size(0);
    ctx->saved_data["xg"] = xg;
    ctx->saved_data["yg"] = yg;
    return mtransform(M, X, false, row_off, col_off, T_out, band_lo, band_hi, xg, yg);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const Tensor M = ctx->get_saved_variables()[0];
    auto& d = ctx->saved_data;
    Tensor dX = mtransform(M, grads[0].contiguous(), true, d["co"].toInt(), d["ro"].toInt(), d["T_in"].toInt(),
                           d["hi"].toInt(), d["lo"].toInt(), d["yg"].toInt(), d["xg"].toInt());
    return {dX, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

struct SpmmFn : public torch::autograd::Function<SpmmFn> {
  // sparse.mm backward: dX_k = Â_kᵀ dY_k (Â is a constant: no gradient, as in the reference)
  static Tensor forward(AutogradContext* ctx, const Tensor& X, const Tensor& rowptr, const Tensor& col,
                        const Tensor& val, const OptTensor& t_rowptr, const OptTensor& t_col, const OptTensor& t_val,
                        int64_t N, double avg, const OptTensor& g_rows, const OptTensor& g_chunks, const OptTensor& t_g_rows,
                        const OptTensor& t_g_chunks, bool need) {
    // `need` is decided by the caller: inside forward() autograd has already switched grad mode off
    at::AutoDispatchBelowADInplaceOrView guard;
    if (need) {
      TORCH_CHECK(t_rowptr.has_value() && t_col.has_value() && t_val.has_value(),
                  "spmm: X requires grad but no transposed adjacency was passed");
      const bool tg = t_g_rows.has_value() && t_g_chunks.has_value();
      ctx->save_for_backward({*t_rowptr, *t_col, *t_val, tg ? *t_g_rows : none_like(X), tg ? *t_g_chunks : none_like(X)});
    }
    ctx->saved_data["N"] = N;
    ctx->saved_data["avg"] = avg;
    return spmm_csr_batched(rowptr, col, val, X, N, avg, g_rows, g_chunks);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto sv = ctx->get_saved_variables();
    const bool tg = sv[3].numel() > 0;
    Tensor dX = spmm_csr_batched(sv[0], sv[1], sv[2], grads[0].contiguous(), ctx->saved_data["N"].toInt(),
                                 ctx->saved_data["avg"].toDouble(), tg ? OptTensor(sv[3]) : OptTensor(), tg ? OptTensor(sv[4]) : OptTensor());
    variable_list out(14);
    out[0] = dX;
    return out;
  }
};

struct FeatureGemmFn : public torch::autograd::Function<FeatureGemmFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& A, const Tensor& W, int64_t act) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto [Y, pre] = bgemm(A, W, false, act, act != TMGCN_ACT_NONE, TMGCN_GEMM_AUTO);
    ctx->saved_data["act"] = act;
    ctx->save_for_backward({A, W, pre});
    return Y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto sv = ctx->get_saved_variables();
    const Tensor &A = sv[0], &W = sv[1], &pre = sv[2];
    const int64_t act = ctx->saved_data["act"].toInt();
    Tensor dY = grads[0].contiguous();
    Tensor dA, dW;
    // layer 1 of the 2-layer models: the input (AtXt / AX) is a constant, so only dW is asked for — the activation
    // gradient is folded into the dW kernel and dY ⊙ act'(pre) never exists as a tensor
    const bool fold_act = act != TMGCN_ACT_NONE && !ctx->needs_input_grad(0) && ctx->needs_input_grad(1) && A.dim() == 3 &&
                          tmgcn_gemm_dw_act_supported((int32_t)A.size(-1), (int32_t)dY.size(-1)) &&
                          reinterpret_cast<uintptr_t>(A.const_data_ptr()) % 8 == 0 &&
                          reinterpret_cast<uintptr_t>(dY.const_data_ptr()) % 8 == 0 &&
                          reinterpret_cast<uintptr_t>(pre.const_data_ptr()) % 8 == 0;
    if (fold_act) {
      dW = bgemm_dW_act(A, dY, pre, act, W.dim() == 3);
      if (dW.scalar_type() != W.scalar_type()) dW = dW.to(W.scalar_type());
      return {dA, dW, Tensor()};
    }
    if (act != TMGCN_ACT_NONE) dY = act_bwd(pre, dY, act);
    if (ctx->needs_input_grad(0)) dA = std::get<0>(bgemm(dY, W, true, TMGCN_ACT_NONE, false, TMGCN_GEMM_AUTO));
    if (ctx->needs_input_grad(1)) {
      dW = bgemm_dW(A, dY, W.dim() == 3, TMGCN_DW_AUTO);  // summed in fp32, rounded once for a bf16 parameter
      if (dW.scalar_type() != W.scalar_type()) dW = dW.to(W.scalar_type());
    }
    return {dA, dW, Tensor()};
  }
};

struct SpmmFeatureGemmFn : public torch::autograd::Function<SpmmFeatureGemmFn> {
  // Fused P2+P3.  Backward uses Âᵀ(dY·Wᵀ) = (Âᵀ·dY)·Wᵀ: the same fused kernel on dY.
  static Tensor forward(AutogradContext* ctx, const Tensor& X, const Tensor& W, const Tensor& rowptr,
                        const Tensor& col, const Tensor& val, const OptTensor& t_rowptr, const OptTensor& t_col,
                        const OptTensor& t_val, int64_t N, double avg, int64_t act, int64_t grid_reserve,
                        const OptTensor& g_rows, const OptTensor& g_chunks, const OptTensor& t_g_rows,
                        const OptTensor& t_g_chunks, bool need_x, bool need_w) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto [Y, AX, pre] = spmm_gemm(rowptr, col, val, X, N, W, false, act, need_w, need_x || need_w, grid_reserve, avg, g_rows, g_chunks);
    if (need_x)
      TORCH_CHECK(t_rowptr.has_value() && t_col.has_value() && t_val.has_value(),
                  "spmm_feature_gemm: X requires grad but no transposed adjacency was passed");
    const bool tg = need_x && t_g_rows.has_value() && t_g_chunks.has_value();
    ctx->save_for_backward({W, AX, pre, need_x ? *t_rowptr : Tensor(), need_x ? *t_col : Tensor(),
                            need_x ? *t_val : Tensor(), tg ? *t_g_rows : none_like(X), tg ? *t_g_chunks : none_like(X)});
    ctx->saved_data["N"] = N;
    ctx->saved_data["avg"] = avg;
    ctx->saved_data["act"] = act;
    ctx->saved_data["reserve"] = grid_reserve;
    return Y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto sv = ctx->get_saved_variables();
    const Tensor &W = sv[0], &AX = sv[1], &pre = sv[2];
    const int64_t N = ctx->saved_data["N"].toInt(), act = ctx->saved_data["act"].toInt();
    Tensor dY = grads[0].contiguous();
    if (act != TMGCN_ACT_NONE) dY = act_bwd(pre, dY, act);
    Tensor dX, dW;
    if (ctx->needs_input_grad(0)) {
      const bool tg = sv[6].numel() > 0;
      const OptTensor gr = tg ? OptTensor(sv[6]) : OptTensor(), gc = tg ? OptTensor(sv[7]) : OptTensor();
      if (spmm_gemm_supported(dY.size(-1), W.size(-2)))
        dX = std::get<0>(spmm_gemm(sv[3], sv[4], sv[5], dY, N, W, true, TMGCN_ACT_NONE, false, false,
                                   ctx->saved_data["reserve"].toInt(), ctx->saved_data["avg"].toDouble(), gr, gc));
      else  // the transposed widths have no fused kernel: dA = dY·Wᵀ, then Âᵀ·dA
        dX = spmm_csr_batched(sv[3], sv[4], sv[5], std::get<0>(bgemm(dY, W, true, TMGCN_ACT_NONE, false, TMGCN_GEMM_AUTO)), N,
                              ctx->saved_data["avg"].toDouble(), gr, gc);
    }
    if (ctx->needs_input_grad(1)) dW = bgemm_dW(AX, dY, W.dim() == 3, TMGCN_DW_AUTO);
    variable_list out(18);
    out[0] = dX;
    out[1] = dW;
    return out;
  }
};

struct EdgeHeadFn : public torch::autograd::Function<EdgeHeadFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& Z, const Tensor& U, const Tensor& src, const Tensor& dst,
                        const OptTensor& eptr, const OptTensor& eidx, bool need) {
    at::AutoDispatchBelowADInplaceOrView guard;
    Tensor Z2 = Z.contiguous().reshape({-1, Z.size(-1)});
    if (need) {
      TORCH_CHECK(eptr.has_value() && eidx.has_value(), "edge_head: gradients needed but no inverted edge index passed");
      ctx->save_for_backward({Z2, U, src, dst, *eptr, *eidx});
    }
    ctx->saved_data["zshape"] = Z.sizes().vec();
    return edge_head_fwd(Z2, src, dst, U);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto sv = ctx->get_saved_variables();
    const bool nz = ctx->needs_input_grad(0), nu = ctx->needs_input_grad(1);
    auto [dZ, dU] = edge_head_bwd(sv[0], sv[2], sv[3], sv[1], grads[0].contiguous(), sv[4], sv[5], nz, nu);
    Tensor gz = nz ? dZ.reshape(ctx->saved_data["zshape"].toIntVector()) : Tensor();
    return {gz, nu ? dU : Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// The constant 1.0 a training step may hand to `loss.backward(gradient=…)` instead of letting autograd fill a fresh
// ones_like(loss) every step: one tensor per device, made once.  HeadLossFn::backward recognises it BY ADDRESS and then
// returns the gradients it formed with upstream gradient 1 unscaled — no scaling launch; any other gradient tensor is
// multiplied in as usual.
static std::mutex g_unit_mutex;
static std::vector<Tensor> g_unit_grad(64);
Tensor unit_gradient(const Tensor& like) {
  TORCH_CHECK(like.is_cuda(), "unit_gradient: a ROCm tensor names the device");
  const int dev = like.device().index();
  std::lock_guard<std::mutex> lock(g_unit_mutex);
  TORCH_CHECK(dev >= 0 && dev < (int)g_unit_grad.size(), "unit_gradient: device index out of range");
  if (!g_unit_grad[dev].defined()) g_unit_grad[dev] = at::ones({}, like.options().dtype(at::kFloat));
  return g_unit_grad[dev];
}
static bool is_unit_gradient(const Tensor& g) {
  if (!g.defined() || !g.is_cuda()) return false;
  const int dev = g.device().index();
  std::lock_guard<std::mutex> lock(g_unit_mutex);
  return dev >= 0 && dev < (int)g_unit_grad.size() && g_unit_grad[dev].defined() && g.numel() == 1 &&
         g.const_data_ptr() == g_unit_grad[dev].const_data_ptr();
}

// loss (and, as a non-differentiable by-product, the logits) of the fused head + criterion.  Two schedules:
//   speculative  the gradients are formed in the SAME launch as the loss (upstream gradient 1) and kept; backward
//                multiplies them by the upstream gradient of the loss in one small launch.  Taken when the entry
//                stream outweighs dZ: link prediction (E >> R), and always when the layer-1 GEMM is folded in
//                (only dW and dU leave the kernel).
//   deferred     forward = the loss-only form (src-side entries only); backward = the gradient form with the upstream
//                gradient as a device scalar inside the kernel.  Taken when dZ is the bigger object (classification:
//                24 k labelled edges among 570 k rows): no pass over dZ just to scale it.
struct HeadLossFn : public torch::autograd::Function<HeadLossFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& Z, const OptTensor& W_fold, const Tensor& U,
                               const Tensor& eptr, const Tensor& arow, const Tensor& ent, const Tensor& other,
                               const Tensor& meta, const Tensor& counts, const Tensor& weight, const Tensor& sync,
                               bool want_logits, bool need, bool unit_grad, const OptTensor& srow, int64_t n_parts) {
    at::AutoDispatchBelowADInplaceOrView guard;
    const bool fold = W_fold.has_value() && W_fold->defined();
    Tensor Z2 = Z.contiguous().reshape({-1, Z.size(-1)});
    const int64_t R = Z2.size(0), F = fold ? W_fold->size(1) : Z2.size(1), E = ent.numel() / 2;
    // unit_grad: the caller will run backward with tmgcn::unit_gradient() — the speculative gradients need no scaling
    // pass then, so one launch does everything whatever the shape
    const bool deferred = need && !fold && !unit_grad && R * F * 8 > E * 10;
    auto [loss, logits, dZ, dU] = head_loss_fwd(Z2, W_fold, U, eptr, arow, ent, other, meta, counts, weight, sync, OptTensor(),
                                                need && !deferred, want_logits, true, srow, n_parts);
    if (need && deferred)
      ctx->save_for_backward({Z2, U, eptr, arow, ent, other, meta, counts, weight, sync,
                              (srow.has_value() && srow->defined()) ? *srow : none_like(Z2)});
    else if (need)
      ctx->save_for_backward({dZ, dU});
    ctx->saved_data["n_parts"] = n_parts;
    ctx->saved_data["zshape"] = Z.sizes().vec();
    ctx->saved_data["fold"] = fold;
    ctx->saved_data["deferred"] = deferred;
    ctx->mark_non_differentiable({logits});
    return {loss, logits};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto sv = ctx->get_saved_variables();
    const bool fold = ctx->saved_data["fold"].toBool(), deferred = ctx->saved_data["deferred"].toBool();
    Tensor g = grads[0].contiguous().to(at::kFloat);
    Tensor gz, gu;
    if (deferred) {
      TORCH_CHECK(sv.size() == 11, "head_loss: backward through a call made without gradients");
      auto out = head_loss_fwd(sv[0], OptTensor(), sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7], sv[8], sv[9], g, true, false,
                               false, sv[10].numel() ? OptTensor(sv[10]) : OptTensor(), ctx->saved_data["n_parts"].toInt());
      gz = std::get<2>(out);
      gu = std::get<3>(out);
    } else {
      TORCH_CHECK(sv.size() == 2, "head_loss: backward through a call made without gradients");
      if (is_unit_gradient(grads[0])) {
        gz = sv[0];
        gu = sv[1];
      } else {
        std::tie(gz, gu) = scale2(g, sv[0], sv[1]);
      }
    }
    variable_list out(16);
    if (fold) out[1] = gz; else out[0] = gz.reshape(ctx->saved_data["zshape"].toIntVector());
    out[2] = gu;
    return out;
  }
};

// Layers 1 + 2 of the narrow 2-layer models (csrc/layer12.hip): Z = act2((Â ⋆ act1(H·W1))·W2) with H a constant.
struct Layer12Fn : public torch::autograd::Function<Layer12Fn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& H, const Tensor& W1, const Tensor& W2, const Tensor& rowptr,
                        const Tensor& col, const Tensor& val, const OptTensor& t_rowptr, const OptTensor& t_col,
                        const OptTensor& t_val, int64_t N, double avg, int64_t act1, int64_t act2, const OptTensor& blk,
                        const OptTensor& t_blk, bool need1, bool need2) {
    at::AutoDispatchBelowADInplaceOrView guard;
    want(H, "layer12 H");
    for (const OptTensor* b : {&blk, &t_blk})
      TORCH_CHECK(!b->has_value() || ((*b)->is_cuda() && (*b)->scalar_type() == at::kLong && (*b)->is_contiguous() && (*b)->dim() == 2 &&
                                      (*b)->size(1) == 2 && (*b)->size(0) >= 1),
                  "layer12: a row-block partition is a contiguous int64 [n, 2] tensor of (first row, rows) pairs on the device");
    want(W1, "layer12 W1");
    want(W2, "layer12 W2");
    check_csr(rowptr, col, val, H, N, "layer12");
    TORCH_CHECK(W1.dim() == 2 && W2.dim() == 2 && W1.size(0) == H.size(2) && W2.size(0) == W1.size(1), "layer12: W1 ",
                W1.sizes(), " / W2 ", W2.sizes(), " do not chain from H ", H.sizes());
    const int64_t K0 = H.size(2), F = W1.size(1), Nf = W2.size(1), R = H.size(0) * H.size(1);
    TORCH_CHECK(tmgcn_layer12_supported((int32_t)K0, (int32_t)F, (int32_t)Nf), "layer12: unsupported widths ", K0, " -> ", F, " -> ", Nf);
    c10::DeviceGuard g(H.device());
    Tensor Z = at::empty({H.size(0), H.size(1), Nf}, H.options());
    Tensor AX = need2 ? at::empty({H.size(0), H.size(1), F}, H.options()) : Tensor();
    Tensor pre2 = ((need1 || need2) && act2 != TMGCN_ACT_NONE) ? at::empty_like(Z) : Tensor();
    // The fused forward re-applies W1 and the non-linearity to every GATHERED row: per non-zero, not per row.  That
    // pays in the entry-major kernel (slices of >= 256 nodes: the work is spread per entry), when a slice is small enough
    // for the staged variant (layer-1 output formed once per node in LDS) and while rows are short; otherwise forming
    // the layer-1 output with the GEMM is cheaper — the same Z to fp32 summation order (tmgcn_layer12_fwd_pays: the
    // library's own measurements), and the fused BACKWARD (which recomputes per row) is used in every case, so no
    // pre-activation is kept.
    if (tmgcn_layer12_fwd_pays(R, (int32_t)N, (int32_t)F, (float)avg)) {
      ok(tmgcn_layer12_fwd_f32((const int64_t*)ptr(rowptr), (const int32_t*)ptr(col), (const float*)ptr(val), (const float*)ptr(H),
                               (const float*)ptr(W1), (int32_t)act1, (const float*)ptr(W2), (int32_t)act2, R, (int32_t)N,
                               (int32_t)K0, (int32_t)F, (int32_t)Nf, (float*)ptr(Z), (float*)ptr(AX), (float*)ptr(pre2), (float)avg,
                               blk.has_value() ? (const int64_t*)ptr(*blk) : nullptr, blk.has_value() ? (int32_t)(blk->numel() / 2) : 0,
                               stream_of(H)),
         "tmgcn_layer12_fwd_f32");
    } else {
      Tensor Y = std::get<0>(bgemm(H, W1, false, act1, false, TMGCN_GEMM_AUTO));
      ok(tmgcn_spmm_gemm_f32_hint((const int64_t*)ptr(rowptr), (const int32_t*)ptr(col), (const float*)ptr(val), (const float*)ptr(Y),
                                  R, (int32_t)N, (int32_t)F, (const float*)ptr(W2), (int32_t)Nf, 0, 0, 0, (int32_t)act2, (float*)ptr(Z),
                                  (float*)ptr(AX), (float*)ptr(pre2), 0, (float)avg, stream_of(H)),
         "tmgcn_spmm_gemm_f32");
    }
    if (need1 || need2) {
      TORCH_CHECK(!need1 || (t_rowptr.has_value() && t_col.has_value() && t_val.has_value()),
                  "layer12: the gradient of W1 needs the transposed adjacency");
      ctx->save_for_backward({H, W1, W2, AX.defined() ? AX : none_like(H), pre2.defined() ? pre2 : none_like(H),
                              need1 ? *t_rowptr : none_like(H), need1 ? *t_col : none_like(H), need1 ? *t_val : none_like(H),
                              (need1 && t_blk.has_value()) ? *t_blk : none_like(H)});
    }
    ctx->saved_data["N"] = N;
    ctx->saved_data["avg"] = avg;
    ctx->saved_data["act1"] = act1;
    ctx->saved_data["act2"] = act2;
    return Z;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto sv = ctx->get_saved_variables();
    TORCH_CHECK(sv.size() == 9, "layer12: backward through a call made without gradients");
    const Tensor &H = sv[0], &W1 = sv[1], &W2 = sv[2], &AX = sv[3], &pre2 = sv[4];
    const int64_t N = ctx->saved_data["N"].toInt(), act1 = ctx->saved_data["act1"].toInt(), act2 = ctx->saved_data["act2"].toInt();
    const double avg = ctx->saved_data["avg"].toDouble();
    Tensor dZ = grads[0].contiguous();
    if (reinterpret_cast<uintptr_t>(dZ.const_data_ptr()) % 8 != 0) dZ = dZ.clone();
    c10::DeviceGuard g(H.device());
    Tensor dW1, dW2;
    const int64_t K0 = H.size(2), F = W1.size(1), Nf = W2.size(1), R = H.size(0) * H.size(1);
    bool dw2_done = false;
    if (ctx->needs_input_grad(1)) {
      dW1 = at::empty_like(W1);
      const int32_t n_blk = sv[8].numel() >= 2 ? (int32_t)(sv[8].numel() / 2) : 0;
      const int64_t need = tmgcn_layer12_bwd_workspace_bytes((int32_t)K0, (int32_t)F, (int32_t)Nf, R, n_blk);
      Tensor ws = at::empty({need}, H.options().dtype(at::kByte));
      // the entry-major backward forms dW2 in the same launch (include/tmgcn.h: tmgcn_layer12_bwd_forms_dw2)
      dw2_done = ctx->needs_input_grad(2) && AX.numel() > 0 &&
                 tmgcn_layer12_bwd_forms_dw2(R, (int32_t)N, (int32_t)F, (int32_t)Nf, (float)avg, n_blk) != 0;
      if (dw2_done) dW2 = at::empty_like(W2);
      ok(tmgcn_layer12_bwd_f32((const int64_t*)ptr(sv[5]), (const int32_t*)ptr(sv[6]), (const float*)ptr(sv[7]), (const float*)ptr(dZ),
                               act2 != TMGCN_ACT_NONE ? (const float*)ptr(pre2) : nullptr, (const float*)ptr(H), (const float*)ptr(W1),
                               (int32_t)act1, (const float*)ptr(W2), (int32_t)act2, R, (int32_t)N, (int32_t)K0, (int32_t)F,
                               (int32_t)Nf, (float*)ptr(dW1), dw2_done ? (const float*)ptr(AX) : nullptr, dw2_done ? (float*)ptr(dW2) : nullptr,
                               (float)avg, n_blk ? (const int64_t*)ptr(sv[8]) : nullptr, n_blk, ptr(ws), ws.numel(), stream_of(H)),
         "tmgcn_layer12_bwd_f32");
    }
    if (ctx->needs_input_grad(2) && !dw2_done) {
      if (act2 != TMGCN_ACT_NONE) dW2 = bgemm_dW_act(AX, dZ, pre2, act2, false);
      else dW2 = bgemm_dW(AX, dZ, false, TMGCN_DW_AUTO);
    }
    variable_list out(17);
    out[1] = dW1;
    out[2] = dW2;
    return out;
  }
};

struct ActivationFn : public torch::autograd::Function<ActivationFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, int64_t act) {
    at::AutoDispatchBelowADInplaceOrView guard;
    ctx->save_for_backward({x});
    ctx->saved_data["act"] = act;
    return act_fwd(x, act);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    return {act_bwd(ctx->get_saved_variables()[0], grads[0].contiguous(), ctx->saved_data["act"].toInt()), Tensor()};
  }
};

struct WeightedCeFn : public torch::autograd::Function<WeightedCeFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& logits, const Tensor& target, const Tensor& weight,
                        int64_t ignore_index) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto [loss, stats] = wce_fwd(logits, target, weight, ignore_index);
    ctx->save_for_backward({logits, target, weight, stats});
    ctx->saved_data["ign"] = ignore_index;
    return loss;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    at::AutoDispatchBelowADInplaceOrView guard;
    auto sv = ctx->get_saved_variables();
    Tensor g = grads[0].contiguous().to(at::kFloat);
    return {wce_bwd(sv[0], sv[1], sv[2], sv[3], g, ctx->saved_data["ign"].toInt()), Tensor(), Tensor(), Tensor()};
  }
};

Tensor m_transform_ad(const Tensor& X, const Tensor& M, int64_t band_lo, int64_t band_hi, int64_t row_off,
                      int64_t col_off, int64_t T_out, int64_t xg, int64_t yg) {
  return MTransformFn::apply(X, M, band_lo, band_hi, row_off, col_off, T_out, xg, yg);
}
Tensor spmm_ad(const Tensor& X, const Tensor& rowptr, const Tensor& col, const Tensor& val, const OptTensor& t_rowptr,
               const OptTensor& t_col, const OptTensor& t_val, int64_t N, double avg, const OptTensor& g_rows,
               const OptTensor& g_chunks, const OptTensor& t_g_rows, const OptTensor& t_g_chunks) {
  return SpmmFn::apply(X, rowptr, col, val, t_rowptr, t_col, t_val, N, avg, g_rows, g_chunks, t_g_rows, t_g_chunks,
                       at::GradMode::is_enabled() && X.requires_grad());
}
Tensor feature_gemm_ad(const Tensor& A, const Tensor& W, int64_t act) { return FeatureGemmFn::apply(A, W, act); }
Tensor spmm_feature_gemm_ad(const Tensor& X, const Tensor& W, const Tensor& rowptr, const Tensor& col,
                            const Tensor& val, const OptTensor& t_rowptr, const OptTensor& t_col,
                            const OptTensor& t_val, int64_t N, double avg, int64_t act, int64_t grid_reserve,
                            const OptTensor& g_rows, const OptTensor& g_chunks, const OptTensor& t_g_rows,
                            const OptTensor& t_g_chunks) {
  const bool grad = at::GradMode::is_enabled();
  return SpmmFeatureGemmFn::apply(X, W, rowptr, col, val, t_rowptr, t_col, t_val, N, avg, act, grid_reserve, g_rows, g_chunks,
                                  t_g_rows, t_g_chunks, grad && X.requires_grad(), grad && W.requires_grad());
}
Tensor edge_head_ad(const Tensor& Z, const Tensor& U, const Tensor& src, const Tensor& dst, const OptTensor& eptr,
                    const OptTensor& eidx) {
  return EdgeHeadFn::apply(Z, U, src, dst, eptr, eidx,
                           at::GradMode::is_enabled() && (Z.requires_grad() || U.requires_grad()));
}
std::tuple<Tensor, Tensor> head_loss_ad(const Tensor& Z, const OptTensor& W_fold, const Tensor& U, const Tensor& eptr,
                                        const Tensor& arow, const Tensor& ent, const Tensor& other, const Tensor& meta,
                                        const Tensor& counts, const Tensor& weight, const Tensor& sync, bool want_logits,
                                        bool unit_grad, const OptTensor& srow, int64_t n_parts) {
  const bool fold = W_fold.has_value() && W_fold->defined();
  const bool need = at::GradMode::is_enabled() && (U.requires_grad() || (fold ? W_fold->requires_grad() : Z.requires_grad()));
  auto out = HeadLossFn::apply(Z, W_fold, U, eptr, arow, ent, other, meta, counts, weight, sync, want_logits, need, unit_grad, srow,
                               n_parts);
  return {out[0], out[1]};
}
Tensor layer12_ad(const Tensor& H, const Tensor& W1, const Tensor& W2, const Tensor& rowptr, const Tensor& col, const Tensor& val,
                  const OptTensor& t_rowptr, const OptTensor& t_col, const OptTensor& t_val, int64_t N, double avg,
                  int64_t act1, int64_t act2, const OptTensor& blk, const OptTensor& t_blk) {
  TORCH_CHECK(!(at::GradMode::is_enabled() && H.requires_grad()), "layer12: H is the model's constant input (no gradient is formed for it)");
  const bool grad = at::GradMode::is_enabled();
  return Layer12Fn::apply(H, W1, W2, rowptr, col, val, t_rowptr, t_col, t_val, N, avg, act1, act2, blk, t_blk,
                          grad && W1.requires_grad(), grad && W2.requires_grad());
}
Tensor activation_ad(const Tensor& x, int64_t act) { return ActivationFn::apply(x, act); }
Tensor weighted_ce_ad(const Tensor& logits, const Tensor& target, const Tensor& weight, int64_t ignore_index) {
  return WeightedCeFn::apply(logits, target, weight, ignore_index);
}

}  // namespace

TORCH_LIBRARY(tmgcn, m) {
  // kernel-level
  m.def("mtransform(Tensor M, Tensor X, bool transpose, int row_off, int col_off, int T_out, int band_lo, int band_hi, "
        "int x_group_rows, int y_group_rows) -> Tensor");
  m.def("mtransform_out(Tensor M, Tensor X, Tensor(a!) Y, bool transpose, int row_off, int col_off, int band_lo, "
        "int band_hi, int x_group_rows, int y_group_rows) -> ()");
  m.def("spmm_csr_batched(Tensor rowptr, Tensor col, Tensor val, Tensor X, int N, float avg_nnz_per_row, "
        "Tensor? giant_rows=None, Tensor? giant_chunks=None) -> Tensor");
  m.def("spmm_gemm(Tensor rowptr, Tensor col, Tensor val, Tensor X, int N, Tensor W, bool trans_w, int act, "
        "bool want_ax, bool want_pre, int grid_reserve, float avg_nnz_per_row=-1.0, Tensor? giant_rows=None, "
        "Tensor? giant_chunks=None) -> (Tensor, Tensor, Tensor)");
  m.def("spmm_gemm_out(Tensor rowptr, Tensor col, Tensor val, Tensor X, int N, Tensor W, bool trans_w, int act, "
        "Tensor(a!) Y, Tensor(b!)? AX, Tensor(c!)? pre, int grid_reserve, float avg_nnz_per_row=-1.0, Tensor? giant_rows=None, "
        "Tensor? giant_chunks=None) -> ()");
  m.def("bgemm(Tensor A, Tensor W, bool trans_w, int act, bool want_pre, int algo) -> (Tensor, Tensor)");
  m.def("bgemm_dW(Tensor A, Tensor dY, bool per_slice, int algo) -> Tensor");
  m.def("bgemm_dW_act(Tensor A, Tensor dY, Tensor pre, int act, bool per_slice) -> Tensor");
  m.def("edge_head_fwd(Tensor Z2, Tensor src, Tensor dst, Tensor U) -> Tensor");
  m.def("edge_head_bwd(Tensor Z2, Tensor src, Tensor dst, Tensor U, Tensor dout, Tensor eptr, Tensor eidx, "
        "bool need_dz, bool need_du) -> (Tensor, Tensor)");
  m.def("act_fwd(Tensor x, int act) -> Tensor");
  m.def("act_bwd(Tensor x, Tensor dy, int act) -> Tensor");
  m.def("wce_fwd(Tensor logits, Tensor target, Tensor weight, int ignore_index) -> (Tensor, Tensor)");
  m.def("wce_bwd(Tensor logits, Tensor target, Tensor weight, Tensor stats, Tensor g, int ignore_index) -> Tensor");
  m.def("head_loss_fwd(Tensor Z, Tensor? W_fold, Tensor U, Tensor eptr, Tensor arow, Tensor ent, Tensor other, Tensor meta, "
        "Tensor counts, Tensor weight, Tensor(a!) sync, Tensor? gscale, bool grad, bool want_logits, bool want_loss, "
        "Tensor? srow=None, int n_parts=0) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def("scale2(Tensor g, Tensor a, Tensor b) -> (Tensor, Tensor)");
  m.def("sgd_step(Tensor(a!)[] params, Tensor[] grads, Tensor(b!)[] bufs, float lr, float momentum, float dampening, "
        "float weight_decay, bool nesterov, bool maximize, bool first_step) -> ()");
  m.def("head_loss_supported(int F, int C, int K) -> bool", &head_loss_supported);
  m.def("spmm_gemm_supported(int K, int Nf) -> bool", &spmm_gemm_supported);
  m.def("edge_head_supported(int F, int C) -> bool", &edge_head_supported);
  m.def("abi_version() -> int", &abi_version);
  // differentiable
  m.def("m_transform(Tensor X, Tensor M, int band_lo, int band_hi, int row_off, int col_off, int T_out, "
        "int x_group_rows, int y_group_rows) -> Tensor");
  m.def("spmm(Tensor X, Tensor rowptr, Tensor col, Tensor val, Tensor? t_rowptr, Tensor? t_col, Tensor? t_val, "
        "int N, float avg_nnz_per_row, Tensor? giant_rows=None, Tensor? giant_chunks=None, Tensor? t_giant_rows=None, "
        "Tensor? t_giant_chunks=None) -> Tensor");
  m.def("feature_gemm(Tensor A, Tensor W, int act) -> Tensor");
  m.def("spmm_feature_gemm(Tensor X, Tensor W, Tensor rowptr, Tensor col, Tensor val, Tensor? t_rowptr, "
        "Tensor? t_col, Tensor? t_val, int N, float avg_nnz_per_row, int act, int grid_reserve, Tensor? giant_rows=None, "
        "Tensor? giant_chunks=None, Tensor? t_giant_rows=None, Tensor? t_giant_chunks=None) -> Tensor");
  m.def("edge_head(Tensor Z, Tensor U, Tensor src, Tensor dst, Tensor? eptr, Tensor? eidx) -> Tensor");
  m.def("activation(Tensor x, int act) -> Tensor");
  m.def("layer12(Tensor H, Tensor W1, Tensor W2, Tensor rowptr, Tensor col, Tensor val, Tensor? t_rowptr, Tensor? t_col, "
        "Tensor? t_val, int N, float avg_nnz_per_row, int act1, int act2, Tensor? row_blocks=None, Tensor? t_row_blocks=None) -> Tensor");
  m.def("layer12_supported(int K0, int F, int Nf) -> bool", &layer12_supported);
  m.def("weighted_ce(Tensor logits, Tensor target, Tensor weight, int ignore_index) -> Tensor");
  m.def("head_loss(Tensor Z, Tensor? W_fold, Tensor U, Tensor eptr, Tensor arow, Tensor ent, Tensor other, Tensor meta, "
        "Tensor counts, Tensor weight, Tensor(a!) sync, bool want_logits, bool unit_grad, Tensor? srow=None, int n_parts=0) -> "
        "(Tensor, Tensor)");
  m.def("head_loss_sgd(Tensor Z, Tensor(a!) W_fold, Tensor(b!) U, Tensor eptr, Tensor arow, Tensor other, Tensor meta, Tensor counts, "
        "Tensor weight, Tensor(c!) sync, Tensor(d!)? buf_W, Tensor(e!)? buf_U, float lr, float momentum, float dampening, "
        "float weight_decay, bool nesterov, bool maximize, bool first_step) -> (Tensor, Tensor, Tensor)");
  m.def("unit_gradient(Tensor like) -> Tensor");
  m.def("widen_params(Tensor[] params) -> Tensor[]");
}

// ROCm tensors carry the CUDA dispatch key in PyTorch-ROCm
TORCH_LIBRARY_IMPL(tmgcn, CUDA, m) {
  m.impl("mtransform", &mtransform);
  m.impl("mtransform_out", &mtransform_out);
  m.impl("spmm_csr_batched", &spmm_csr_batched);
  m.impl("spmm_gemm", &spmm_gemm);
  m.impl("spmm_gemm_out", &spmm_gemm_out);
  m.impl("bgemm", &bgemm);
  m.impl("bgemm_dW", &bgemm_dW);
  m.impl("bgemm_dW_act", &bgemm_dW_act);
  m.impl("edge_head_fwd", &edge_head_fwd);
  m.impl("edge_head_bwd", &edge_head_bwd);
  m.impl("act_fwd", &act_fwd);
  m.impl("act_bwd", &act_bwd);
  m.impl("wce_fwd", &wce_fwd);
  m.impl("wce_bwd", &wce_bwd);
  m.impl("head_loss_fwd", &head_loss_fwd);
  m.impl("head_loss_sgd", &head_loss_sgd);
  m.impl("scale2", &scale2);
  m.impl("sgd_step", &sgd_step);
  m.impl("unit_gradient", &unit_gradient);
  // below the Autograd key (inference mode, or called from inside another autograd node) the
  // differentiable operators are their plain forwards
  m.impl("m_transform", &m_transform_ad);
  m.impl("spmm", &spmm_ad);
  m.impl("feature_gemm", &feature_gemm_ad);
  m.impl("spmm_feature_gemm", &spmm_feature_gemm_ad);
  m.impl("edge_head", &edge_head_ad);
  m.impl("activation", &activation_ad);
  m.impl("layer12", &layer12_ad);
  m.impl("widen_params", &widen_params_ad);
  m.impl("weighted_ce", &weighted_ce_ad);
  m.impl("head_loss", &head_loss_ad);
}

TORCH_LIBRARY_IMPL(tmgcn, Autograd, m) {
  m.impl("m_transform", &m_transform_ad);
  m.impl("spmm", &spmm_ad);
  m.impl("feature_gemm", &feature_gemm_ad);
  m.impl("spmm_feature_gemm", &spmm_feature_gemm_ad);
  m.impl("edge_head", &edge_head_ad);
  m.impl("activation", &activation_ad);
  m.impl("layer12", &layer12_ad);
  m.impl("widen_params", &widen_params_ad);
  m.impl("weighted_ce", &weighted_ce_ad);
  m.impl("head_loss", &head_loss_ad);
}

// a CPU tensor reaching a kernel-level op gets the reference-style RuntimeError, not "no kernel"
TORCH_LIBRARY_IMPL(tmgcn, CPU, m) {
  m.impl("mtransform", &mtransform);
  m.impl("spmm_csr_batched", &spmm_csr_batched);
  m.impl("spmm_gemm", &spmm_gemm);
  m.impl("bgemm", &bgemm);
  m.impl("bgemm_dW", &bgemm_dW);
  m.impl("edge_head_fwd", &edge_head_fwd);
  m.impl("act_fwd", &act_fwd);
  m.impl("act_bwd", &act_bwd);
  m.impl("m_transform", &m_transform_ad);
  m.impl("spmm", &spmm_ad);
  m.impl("feature_gemm", &feature_gemm_ad);
  m.impl("spmm_feature_gemm", &spmm_feature_gemm_ad);
  m.impl("edge_head", &edge_head_ad);
  m.impl("activation", &activation_ad);
  m.impl("layer12", &layer12_ad);
  m.impl("widen_params", &widen_params_ad);
  m.impl("weighted_ce", &weighted_ce_ad);
  m.impl("head_loss", &head_loss_ad);
}

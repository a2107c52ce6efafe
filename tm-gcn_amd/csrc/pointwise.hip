// P5 — pointwise non-linearity between layers (embedding_help_functions.py:284-289, 332-334,
// 486) and the library's error plumbing.  Pure HBM streams: 16 B per lane.
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace tmgcn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x,
                                                       float* __restrict__ y, int64_t n, int act,
                                                       int vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const int64_t n4 = n / 4;
    for (int64_t q = i; q < n4; q += stride) {
      float4 v = reinterpret_cast<const float4*>(x)[q];
      v.x = act_apply(v.x, act);
      v.y = act_apply(v.y, act);
      v.z = act_apply(v.z, act);
      v.w = act_apply(v.w, act);
      reinterpret_cast<float4*>(y)[q] = v;
    }
    for (int64_t q = n4 * 4 + i; q < n; q += stride) y[q] = act_apply(x[q], act);
  } else {
    for (int64_t q = i; q < n; q += stride) y[q] = act_apply(x[q], act);
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ dy,
                                                       float* __restrict__ dx, int64_t n, int act,
                                                       int vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const int64_t n4 = n / 4;
    for (int64_t q = i; q < n4; q += stride) {
      const float4 v = reinterpret_cast<const float4*>(x)[q];
      float4 g = reinterpret_cast<const float4*>(dy)[q];
      g.x *= act_grad(v.x, act);
      g.y *= act_grad(v.y, act);
      g.z *= act_grad(v.z, act);
      g.w *= act_grad(v.w, act);
      reinterpret_cast<float4*>(dx)[q] = g;
    }
    for (int64_t q = n4 * 4 + i; q < n; q += stride) dx[q] = dy[q] * act_grad(x[q], act);
  } else {
    for (int64_t q = i; q < n; q += stride) dx[q] = dy[q] * act_grad(x[q], act);
  }
}

// One launch for the optimizer step of every parameter (the scripts' t.optim.SGD(gcn.parameters(), lr, momentum):
// experiment_reddit_our_link_prediction.py:68, 80) — torch's foreach implementation is three to four launches plus
// a zero-fill per parameter list.  fp32 arithmetic; bf16-stored tensors are widened on load and rounded once.
constexpr int kSgdMaxTensors = 16;
struct SgdArgs {
  void* param[kSgdMaxTensors];
  const void* grad[kSgdMaxTensors];
  void* buf[kSgdMaxTensors];
  int32_t first_block[kSgdMaxTensors + 1];
  int32_t numel[kSgdMaxTensors];
  int32_t n, bf16;
  float lr, momentum, dampening, weight_decay;
  int32_t nesterov, maximize, first_step;
};

__device__ __forceinline__ float ld_elem(const void* p, int i, int bf16) {
  if (bf16) return __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(p)[i] << 16);
  return reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_elem(void* p, int i, int bf16, float v) {
  if (bf16) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);                  // round to nearest even (NaN payloads aside: weights are finite)
    reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)(u >> 16);
  } else {
    reinterpret_cast<float*>(p)[i] = v;
  }
}

__global__ __launch_bounds__(256) void sgd_step_kernel(SgdArgs a) {
  int t = 0;
#pragma unroll
  for (int k = 1; k < kSgdMaxTensors; ++k)
    if (k < a.n && (int)blockIdx.x >= a.first_block[k]) t = k;
  const int i = ((int)blockIdx.x - a.first_block[t]) * 256 + threadIdx.x;
  if (i >= a.numel[t]) return;
  float p = ld_elem(a.param[t], i, a.bf16);
  float g = ld_elem(a.grad[t], i, a.bf16);
  if (a.maximize) g = -g;
  if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
  if (a.momentum != 0.f) {
    float b;
    if (a.first_step) b = g;
    else b = __fmul_rn(a.momentum, ld_elem(a.buf[t], i, a.bf16)) + (1.f - a.dampening) * g;   // buf.mul_(m).add_(g, alpha=1-d)
    st_elem(a.buf[t], i, a.bf16, b);
    g = a.nesterov ? fmaf(a.momentum, b, g) : b;
  }
  st_elem(a.param[t], i, a.bf16, fmaf(-a.lr, g, p));
}

// bf16 <-> fp32 of up to 16 small tensors in one launch (the "bf16 weights" configuration widens its two or three
// parameters every step and rounds their gradients back: six single-tensor cast launches otherwise)
struct CastArgs {
  const void* src[kSgdMaxTensors];
  void* dst[kSgdMaxTensors];
  int32_t first_block[kSgdMaxTensors + 1];
  int32_t numel[kSgdMaxTensors];
  int32_t n, to_bf16;
};

__global__ __launch_bounds__(256) void cast_multi_kernel(CastArgs a) {
  int t = 0;
#pragma unroll
  for (int k = 1; k < kSgdMaxTensors; ++k)
    if (k < a.n && (int)blockIdx.x >= a.first_block[k]) t = k;
  const int i = ((int)blockIdx.x - a.first_block[t]) * 256 + threadIdx.x;
  if (i >= a.numel[t]) return;
  if (a.to_bf16) st_elem(a.dst[t], i, 1, reinterpret_cast<const float*>(a.src[t])[i]);
  else reinterpret_cast<float*>(a.dst[t])[i] = ld_elem(a.src[t], i, 1);
}

static unsigned stream_grid(int64_t n) {
  int64_t b = (n / 4 + 255) / 256;
  if (b < 1) b = 1;
  if (b > 256 * 8) b = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  return (unsigned)b;
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_abi_version(void) { return 5; }

extern "C" const char* tmgcn_last_error(void) { return g_err; }

extern "C" int tmgcn_act_fwd_f32(const float* x, float* y, int64_t n, int32_t act, void* stream) {
  TMGCN_REQUIRE(n >= 0, "act_fwd: negative length");
  TMGCN_REQUIRE(act >= TMGCN_ACT_NONE && act <= TMGCN_ACT_SELU, "act_fwd: unknown activation %d", act);
  if (n == 0) return TMGCN_OK;
  TMGCN_REQUIRE(x && y, "act_fwd: null pointer");
  const int vec = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n,
                     act, vec);
  return check_launch("act_fwd");
}

extern "C" int tmgcn_act_bwd_f32(const float* x, const float* dy, float* dx, int64_t n, int32_t act,
                                  void* stream) {
  TMGCN_REQUIRE(n >= 0, "act_bwd: negative length");
  TMGCN_REQUIRE(act >= TMGCN_ACT_NONE && act <= TMGCN_ACT_SELU, "act_bwd: unknown activation %d", act);
  if (n == 0) return TMGCN_OK;
  TMGCN_REQUIRE(x && dy && dx, "act_bwd: null pointer");
  const int vec = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(dy) % 16 == 0) &&
                  (reinterpret_cast<uintptr_t>(dx) % 16 == 0);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx,
                     n, act, vec);
  return check_launch("act_bwd");
}

extern "C" int tmgcn_sgd_step(void* const* params, const void* const* grads, void* const* momentum_bufs, const int64_t* numel,
                              int32_t n, int32_t bf16, float lr, float momentum, float dampening, float weight_decay,
                              int32_t nesterov, int32_t maximize, int32_t first_step, void* stream) {
  TMGCN_REQUIRE(n >= 0 && n <= kSgdMaxTensors, "sgd_step: 0 <= n <= %d tensors per call (got %d)", kSgdMaxTensors, n);
  if (n == 0) return TMGCN_OK;
  TMGCN_REQUIRE(params && grads && numel && (momentum == 0.f || momentum_bufs), "sgd_step: null pointer");
  SgdArgs a{};
  int64_t blocks = 0;
  for (int k = 0; k < n; ++k) {
    TMGCN_REQUIRE(params[k] && grads[k] && numel[k] >= 0 && numel[k] < (int64_t)0x7fffffff && (momentum == 0.f || momentum_bufs[k]),
                  "sgd_step: tensor %d: null pointer or bad size", k);
    a.param[k] = params[k];
    a.grad[k] = grads[k];
    a.buf[k] = momentum != 0.f ? momentum_bufs[k] : nullptr;
    a.numel[k] = (int32_t)numel[k];
    a.first_block[k] = (int32_t)blocks;
    blocks += (numel[k] + 255) / 256;
    TMGCN_REQUIRE(blocks < (int64_t)0x7fffffff, "sgd_step: too many elements for one launch");
  }
  a.first_block[n] = (int32_t)blocks;
  a.n = n;
  a.bf16 = bf16;
  a.lr = lr;
  a.momentum = momentum;
  a.dampening = dampening;
  a.weight_decay = weight_decay;
  a.nesterov = nesterov;
  a.maximize = maximize;
  a.first_step = first_step;
  if (blocks == 0) return TMGCN_OK;
  hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("sgd_step");
}

extern "C" int tmgcn_cast_multi(const void* const* src, void* const* dst, const int64_t* numel, int32_t n, int32_t to_bf16,
                                void* stream) {
  TMGCN_REQUIRE(n >= 0 && n <= kSgdMaxTensors, "cast_multi: 0 <= n <= %d tensors per call (got %d)", kSgdMaxTensors, n);
  if (n == 0) return TMGCN_OK;
  TMGCN_REQUIRE(src && dst && numel, "cast_multi: null pointer");
  CastArgs a{};
  int64_t blocks = 0;
  for (int k = 0; k < n; ++k) {
    TMGCN_REQUIRE(numel[k] >= 0 && numel[k] < (int64_t)0x7fffffff && (numel[k] == 0 || (src[k] && dst[k])),
                  "cast_multi: tensor %d: null pointer or bad size", k);
    a.src[k] = src[k];
    a.dst[k] = dst[k];
    a.numel[k] = (int32_t)numel[k];
    a.first_block[k] = (int32_t)blocks;
    blocks += (numel[k] + 255) / 256;
    TMGCN_REQUIRE(blocks < (int64_t)0x7fffffff, "cast_multi: too many elements for one launch");
  }
  a.first_block[n] = (int32_t)blocks;
  a.n = n;
  a.to_bf16 = to_bf16;
  if (blocks == 0) return TMGCN_OK;
  hipLaunchKernelGGL(cast_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("cast_multi");
}

// P5 — pointwise non-linearity between layers (embedding_help_functions.py:284-289, 332-334,
// 486) and the library's error plumbing.  Pure HBM streams: 16 B per lane.
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include "common.h"

namespace tmgcn {

static thread_local char g_err[512] = "";

constexpr int kCounterPool = 4096;  // 64 groups of 64: launches in flight on different streams never share a group
constexpr int kMaxDevices = 16;
__device__ unsigned int g_tile_counters[kCounterPool];


unsigned int* acquire_tile_counters(hipStream_t stream, int n) {
  if (n < 1 || n > 64) return nullptr;
  static unsigned int* base[kMaxDevices] = {nullptr};
  static std::atomic<unsigned> next{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= kMaxDevices) dev = 0;
  if (!base[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tile_counters)) != hipSuccess) return nullptr;
    base[dev] = static_cast<unsigned int*>(p);
  }
  // slots are handed out in aligned groups of 64 so that n consecutive counters never wrap
  unsigned int* c = base[dev] + (next.fetch_add(1) % (kCounterPool / 64)) * 64;
  if (hipMemsetAsync(c, 0, sizeof(unsigned int) * n, stream) != hipSuccess) return nullptr;
  return c;
}

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

static unsigned stream_grid(int64_t n) {
  int64_t b = (n / 4 + 255) / 256;
  if (b < 1) b = 1;
  if (b > 256 * 8) b = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  return (unsigned)b;
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_abi_version(void) { return 4; }

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

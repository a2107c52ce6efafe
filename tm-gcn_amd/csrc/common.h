// Shared helpers for the gfx950 kernels of the TM-GCN layer (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "tmgcn.h"

namespace tmgcn {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return TMGCN_ERR_LAUNCH;
  }
  return TMGCN_OK;
}

#define TMGCN_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ::tmgcn::set_error(__VA_ARGS__);      \
      return TMGCN_ERR_INVALID;             \
    }                                       \
  } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

__device__ __forceinline__ float act_apply(float x, int act) {
  // torch.nn.ReLU / LeakyReLU(0.01) / SELU constants (ehf:284-289)
  switch (act) {
    case TMGCN_ACT_RELU: return x > 0.f ? x : 0.f;
    case TMGCN_ACT_LEAKY: return x > 0.f ? x : 0.01f * x;
    case TMGCN_ACT_SELU: {
      const float scale = 1.0507009873554804934193349852946f;
      const float alpha = 1.6732632423543772848170429916717f;
      return x > 0.f ? scale * x : scale * alpha * (expf(x) - 1.f);
    }
    default: return x;
  }
}

__device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case TMGCN_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case TMGCN_ACT_LEAKY: return x > 0.f ? 1.f : 0.01f;
    case TMGCN_ACT_SELU: {
      const float scale = 1.0507009873554804934193349852946f;
      const float alpha = 1.6732632423543772848170429916717f;
      return x > 0.f ? scale : scale * alpha * expf(x);
    }
    default: return 1.f;
  }
}

}  // namespace tmgcn

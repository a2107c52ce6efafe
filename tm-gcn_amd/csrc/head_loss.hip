// One-pass edge head + class-weighted cross entropy + every gradient of both   (gfx950 / CDNA4)
//
// Replaces, for the narrow heads of the reference's own experiments (even F <= 8, C <= 4), the chain
//     gather / cat / ·U                      embedding_help_functions.py:228-232, 351-355, 491-495
//     nn.CrossEntropyLoss(weight=w)          experiment_reddit_our_link_prediction.py:69, 79
//     autograd through both                  (index backward, catᵀ·dout, softmax − onehot)
// which ran as five edge-sized launches (head forward, loss forward, loss backward, dU, dZ) with the
// logits and their gradient round-tripping through memory between them (E = 3.2 M labelled edges at
// the Reddit-LP shape: 20x the real edges), plus three reduction tails.
//
// ROW-centric, not edge-centric: the kernel walks the INVERTED edge index (the entries of row r are
// the labelled edges r is an endpoint of, 2·edge + role, ascending — built once per edge set, already
// used by the atomic-free dZ kernel).  A group of G lanes owns a row; a lane takes entries gl, gl+G, …:
// it reads the entry stream (entry id, row of the OTHER endpoint, target class: 9 bytes, coalesced),
// gathers the other endpoint's row (the group's own row sits in registers), recomputes that edge's
// logits — an edge is seen from both of its endpoints, 2·F·C fmas each time, nothing next to the
// gather — and forms  g = w[t]·(softmax(z) − onehot(t)).  Summed over the row's entries BEFORE the
// product with U (as the standalone dZ kernel does) that is the row's dZ, written once; no dlogits
// array exists, nothing is scattered, no atomics.  The entry that sees an edge from its src side also
// adds the edge's loss term and its dU contribution (per-lane fp64 accumulators) and stores the
// logits if the caller wants them.  Σ_e w[t_e] depends only on the targets: the per-class edge counts
// come with the plan, so the gradient needs no second pass.
//
// Tail: one slab of fp64 partials per block; the LAST block to finish (device counter that it resets)
// adds the slabs in a fixed order and writes loss, dU (and dW).  One launch, bitwise reproducible.
//
// K > 0 ("fold"): the 1-layer model's  Z = AtXt·W  (ehf:222, F0 = K = 2) is recomputed on the fly from
// the 8-byte AtXt rows — same fmaf chain as the standalone small GEMM, so the same bits — and dW =
// Σ_r AtXt[r]ᵀ·dZ[r] is accumulated in the same pass: Z and dZ are never stored, and the whole
// training epoch of EmbeddingGCN (cached AtXt) is this one kernel plus the optimizer step.
#include "common.h"

namespace tmgcn {

struct HeadLossArgs {
  const float* Z;              // [R][F]   (fold: AtXt [R][K])
  const float* Wf;             // fold: [K][F]
  const float* U;              // [2F][C]
  const int32_t* eptr;         // [R+1]
  const int32_t* ent;          // [2E]  2*edge + role
  const int32_t* other;        // [2E]  row of the other endpoint
  const uint8_t* tgt;          // [2E]  target class of the entry's edge, 255 = ignored
  const int64_t* class_count;  // [C]
  const float* weight;         // [C]
  float* logits;               // [E][C] or null
  float* dZ;                   // [R][F] or null
  double* part;                // [blocks][NP]
  float* loss;
  float* dU;
  float* dW;
  int32_t* sync;
  int64_t R;
  int32_t logG;
};

// acc[i] over the block: xor butterfly per wave, the four waves in order through LDS.  Thread t < NP
// returns the block total of acc[t] (others return 0).  Fixed order: reproducible.
template <int NP>
__device__ __forceinline__ double block_reduce(const double (&acc)[NP], double (*red)[NP]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    double v = acc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  double total = 0.0;
  if (threadIdx.x < NP) total = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  __syncthreads();
  return total;
}

template <int FT, int K>
__device__ __forceinline__ void load_row(const HeadLossArgs& a, int64_t r, float (&z)[FT], float (&x)[K ? K : 1]) {
  if constexpr (K == 0) {
    const float2* p = reinterpret_cast<const float2*>(a.Z + r * FT);
#pragma unroll
    for (int i = 0; i < FT / 2; ++i) {
      const float2 v = p[i];
      z[2 * i] = v.x;
      z[2 * i + 1] = v.y;
    }
  } else {
    static_assert(K == 0 || K == 2, "fold supports the reference's F0 = 2");
    const float2 v = *reinterpret_cast<const float2*>(a.Z + r * K);
    x[0] = v.x;
    x[1] = v.y;
    const float* __restrict__ W = a.Wf;
#pragma unroll
    for (int f = 0; f < FT; ++f) z[f] = fmaf(x[1], W[FT + f], fmaf(x[0], W[f], 0.f));   // gemm_small's chain
  }
}

// GRAD: also dZ (or dW when folding) and dU.  Without it only role-0 entries are visited (loss, logits).
template <int FT, int CT, bool GRAD, int K>
__global__ __launch_bounds__(256) void head_loss_small_kernel(HeadLossArgs a) {
  constexpr int NO = 2 * FT * CT, NW = K * FT;
  constexpr int NP = 1 + (GRAD ? NO + NW : 0);      // slab: Σ w·nll | dU [2F][C] | dW [K][F]
  __shared__ double red[4][NP];
  __shared__ int is_last;
  const float* __restrict__ U = a.U;
  float w[CT];
  double den = 0.0;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    w[c] = a.weight[c];
    den += (double)a.class_count[c] * (double)w[c];
  }
  const double invden = 1.0 / den;                   // no labelled edge with weight: 0/0 = NaN, as torch
  const int G = 1 << a.logG;
  const int gl = threadIdx.x & (G - 1);
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ngroups = ((int64_t)gridDim.x * 256) >> a.logG;
  double acc[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) acc[i] = 0.0;

  for (int64_t r = tid >> a.logG; r < a.R; r += ngroups) {
    float zo[FT], xo[K ? K : 1];
    load_row<FT, K>(a, r, zo, xo);
    double S[2][CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) S[0][c] = S[1][c] = 0.0;
    const int p1 = a.eptr[r + 1];
    for (int p = a.eptr[r] + gl; p < p1; p += G) {
      const int x = a.ent[p];
      const bool role = x & 1;
      if (!GRAD && role) continue;
      const int t = a.tgt[p];
      float zt[FT], xt[K ? K : 1];
      load_row<FT, K>(a, a.other[p], zt, xt);
      // logits: f ascending, src then dst — the order of edge_head_fwd_small, from either endpoint
      float lg[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) lg[c] = 0.f;
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const float sv = role ? zt[f] : zo[f], dv = role ? zo[f] : zt[f];
#pragma unroll
        for (int c = 0; c < CT; ++c) lg[c] = fmaf(dv, U[(FT + f) * CT + c], fmaf(sv, U[f * CT + c], lg[c]));
      }
      float mx = lg[0];
#pragma unroll
      for (int c = 1; c < CT; ++c) mx = fmaxf(mx, lg[c]);
      float ex[CT], s = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        ex[c] = expf(lg[c] - mx);
        s += ex[c];
      }
      const bool valid = t < CT;
      float wt = 0.f, zt_t = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c == t) {
          wt = w[c];
          zt_t = lg[c];
        }
      float g[CT];
      const float inv = 1.f / s;
#pragma unroll
      for (int c = 0; c < CT; ++c) g[c] = valid ? wt * (ex[c] * inv - (c == t ? 1.f : 0.f)) : 0.f;
      if constexpr (GRAD) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          if (role) S[1][c] += (double)g[c]; else S[0][c] += (double)g[c];
        }
      }
      if (!role) {
        if (valid) acc[0] += (double)wt * ((double)mx + (double)logf(s) - (double)zt_t);
        if (a.logits) {
          float* o = a.logits + (int64_t)(x >> 1) * CT;
          if constexpr (CT == 2) {
            *reinterpret_cast<float2*>(o) = make_float2(lg[0], lg[1]);
          } else if constexpr (CT == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(lg[0], lg[1], lg[2], lg[3]);
          } else {
#pragma unroll
            for (int c = 0; c < CT; ++c) o[c] = lg[c];
          }
        }
        if constexpr (GRAD) {                       // dU += [z_src, z_dst]ᵀ · g   (this lane is the src side)
#pragma unroll
          for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              acc[1 + f * CT + c] = fma((double)zo[f], (double)g[c], acc[1 + f * CT + c]);
              acc[1 + (FT + f) * CT + c] = fma((double)zt[f], (double)g[c], acc[1 + (FT + f) * CT + c]);
            }
        }
      }
    }
    if constexpr (GRAD) {
      for (int o = G >> 1; o > 0; o >>= 1) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          S[0][c] += __shfl_xor(S[0][c], o);
          S[1][c] += __shfl_xor(S[1][c], o);
        }
      }
      float s0[CT], s1[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        s0[c] = (float)(S[0][c] * invden);
        s1[c] = (float)(S[1][c] * invden);
      }
      float dz[FT];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) v = fmaf(s1[c], U[(FT + f) * CT + c], fmaf(s0[c], U[f * CT + c], v));
        dz[f] = v;
      }
      if constexpr (K == 0) {
        float2* o = reinterpret_cast<float2*>(a.dZ + r * FT);
        for (int i = gl; i < FT / 2; i += G) {
          float2 v;
#pragma unroll
          for (int q = 0; q < FT / 2; ++q)
            if (q == i) v = make_float2(dz[2 * q], dz[2 * q + 1]);
          o[i] = v;
        }
      } else if (gl == 0) {                         // dW += AtXt[r]ᵀ · dZ[r]  (dz already carries 1/Σw)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
          for (int f = 0; f < FT; ++f)
            acc[1 + NO + k * FT + f] = fma((double)xo[k], (double)dz[f], acc[1 + NO + k * FT + f]);
      }
    }
  }

  const double mine = block_reduce<NP>(acc, red);
  if (threadIdx.x < NP) a.part[(int64_t)blockIdx.x * NP + threadIdx.x] = mine;
  // hand-off to the last block (cdna_hip_programming.md, in-launch split-K reduction): every storing wave
  // drains its stores, the block meets, ONE lane publishes with an agent-scope release and takes a ticket;
  // the block that draws the last ticket acquires once and then reads the slabs with plain loads.  The
  // XCDs' L2s are not coherent with each other: anything weaker returns stale slabs.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = __hip_atomic_fetch_add(a.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = ticket == (int)gridDim.x - 1;
    if (is_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (!is_last) return;
  // the last block: thread t adds slabs t, t+256, … of every output, then the same block reduction
#pragma unroll
  for (int i = 0; i < NP; ++i) acc[i] = 0.0;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) {
    const double* P = a.part + (int64_t)b * NP;
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] += P[i];
  }
  const double total = block_reduce<NP>(acc, red);
  if (threadIdx.x == 0) {
    a.loss[0] = (float)(total * invden);
    *a.sync = 0;                                    // ready for the next launch: no memset node per step
  }
  if constexpr (GRAD) {
    if (threadIdx.x >= 1 && threadIdx.x < 1 + NO) a.dU[threadIdx.x - 1] = (float)(total * invden);
    if (NW && threadIdx.x >= 1 + NO && threadIdx.x < NP) a.dW[threadIdx.x - 1 - NO] = (float)total;
  }
}

constexpr int kHeadLossMaxBlocks = 1024;

static int head_loss_logG(int64_t E, int64_t R) {
  const int64_t avg = R > 0 ? (2 * E + R - 1) / R : 0;
  int lg = 0;
  while (lg < 5 && ((int64_t)2 << lg) < avg) ++lg;   // chains of about two entries per lane, G <= 32
  return lg;
}

static int head_loss_blocks(int64_t R, int logG) {
  int64_t b = ((R << logG) + 255) / 256;
  if (b > kHeadLossMaxBlocks) b = kHeadLossMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

template <int FT, int CT>
static void head_loss_launch(const HeadLossArgs& a, bool grad, int K, unsigned blocks, hipStream_t st) {
  if (K == 0) {
    if (grad) hipLaunchKernelGGL((head_loss_small_kernel<FT, CT, true, 0>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((head_loss_small_kernel<FT, CT, false, 0>), dim3(blocks), dim3(256), 0, st, a);
  } else {
    if (grad) hipLaunchKernelGGL((head_loss_small_kernel<FT, CT, true, 2>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((head_loss_small_kernel<FT, CT, false, 2>), dim3(blocks), dim3(256), 0, st, a);
  }
}

template <int FT>
static void head_loss_launch_c(const HeadLossArgs& a, int C, bool grad, int K, unsigned blocks, hipStream_t st) {
  switch (C) {
    case 1: head_loss_launch<FT, 1>(a, grad, K, blocks, st); break;
    case 2: head_loss_launch<FT, 2>(a, grad, K, blocks, st); break;
    case 3: head_loss_launch<FT, 3>(a, grad, K, blocks, st); break;
    default: head_loss_launch<FT, 4>(a, grad, K, blocks, st);
  }
}

// dst_a = g·a, dst_b = g·b in one launch (g a device scalar: the upstream gradient of the loss)
__global__ __launch_bounds__(256) void scale2_kernel(const float* __restrict__ g, const float* __restrict__ a,
                                                      float* __restrict__ oa, int64_t na, const float* __restrict__ b,
                                                      float* __restrict__ ob, int64_t nb) {
  const float s = g[0];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += stride) {
    if (i < na) oa[i] = s * a[i]; else ob[i - na] = s * b[i - na];
  }
}

}  // namespace tmgcn

using namespace tmgcn;

extern "C" int tmgcn_head_loss_supported(int32_t F, int32_t C, int32_t K) {
  return (F >= 2 && F <= 8 && F % 2 == 0 && C >= 1 && C <= 4 && (K == 0 || K == 2)) ? 1 : 0;
}

extern "C" int64_t tmgcn_head_loss_workspace_bytes(int32_t F, int32_t C, int32_t K) {
  if (!tmgcn_head_loss_supported(F, C, K)) return 0;
  return (int64_t)kHeadLossMaxBlocks * (1 + 2 * F * C + K * F) * (int64_t)sizeof(double);
}

extern "C" int tmgcn_head_loss_f32(const float* Z, const float* W_fold, int32_t K, const float* U,
                                    const int32_t* eptr, const int32_t* ent, const int32_t* other, const uint8_t* tgt,
                                    const int64_t* class_count, const float* weight, int64_t R, int64_t E, int32_t F,
                                    int32_t C, float* logits, float* loss, float* dZ, float* dU, float* dW,
                                    void* workspace, int64_t workspace_bytes, int32_t* sync, void* stream) {
  TMGCN_REQUIRE(tmgcn_head_loss_supported(F, C, K), "head_loss: unsupported widths F=%d C=%d K=%d (even F <= 8, C <= 4, K in {0, 2})",
                F, C, K);
  TMGCN_REQUIRE(R > 0 && E > 0 && R < (int64_t)0x7fffffff && 2 * E < (int64_t)0x7fffffff,
                "head_loss: need 0 < R, 2E < 2^31 (got R=%lld E=%lld)", (long long)R, (long long)E);
  TMGCN_REQUIRE(Z && U && eptr && ent && other && tgt && class_count && weight && loss && sync && workspace,
                "head_loss: null pointer");
  TMGCN_REQUIRE((K == 0) == (W_fold == nullptr), "head_loss: W_fold must be given exactly when K > 0");
  const bool grad = dU != nullptr;
  TMGCN_REQUIRE(!grad || (K ? dW != nullptr : dZ != nullptr), "head_loss: gradients asked for (dU) but no %s", K ? "dW" : "dZ");
  TMGCN_REQUIRE(reinterpret_cast<uintptr_t>(Z) % 8 == 0 && (!dZ || reinterpret_cast<uintptr_t>(dZ) % 8 == 0) &&
                    (!logits || reinterpret_cast<uintptr_t>(logits) % 16 == 0),
                "head_loss: Z / dZ must be 8-byte aligned, logits 16-byte aligned");
  if (workspace_bytes < tmgcn_head_loss_workspace_bytes(F, C, K)) {
    set_error("head_loss: workspace %lld B < required %lld B", (long long)workspace_bytes,
              (long long)tmgcn_head_loss_workspace_bytes(F, C, K));
    return TMGCN_ERR_WORKSPACE;
  }
  HeadLossArgs a{Z, W_fold, U, eptr, ent, other, tgt, class_count, weight, logits, dZ, (double*)workspace,
                 loss, dU, dW, sync, R, head_loss_logG(E, R)};
  const unsigned blocks = (unsigned)head_loss_blocks(R, a.logG);
  hipStream_t st = (hipStream_t)stream;
  switch (F) {
    case 2: head_loss_launch_c<2>(a, C, grad, K, blocks, st); break;
    case 4: head_loss_launch_c<4>(a, C, grad, K, blocks, st); break;
    case 6: head_loss_launch_c<6>(a, C, grad, K, blocks, st); break;
    default: head_loss_launch_c<8>(a, C, grad, K, blocks, st);
  }
  return check_launch("head_loss");
}

extern "C" int tmgcn_scale2_f32(const float* g, const float* a, float* out_a, int64_t na, const float* b, float* out_b,
                                 int64_t nb, void* stream) {
  TMGCN_REQUIRE(g && na >= 0 && nb >= 0 && (na == 0 || (a && out_a)) && (nb == 0 || (b && out_b)), "scale2: bad arguments");
  if (na + nb == 0) return TMGCN_OK;
  int64_t blocks = (na + nb + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, a, out_a, na, b, out_b, nb);
  return check_launch("scale2");
}

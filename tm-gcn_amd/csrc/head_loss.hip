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
// the labelled edges r is an endpoint of, ascending — built once per edge set, already used by the
// atomic-free dZ kernel) over the list of ACTIVE rows (rows with at least one entry: classification
// configs label 24 k edges among 570 k rows).  A group of G lanes owns a row; a lane takes entries
// gl, gl+G, …: it reads the entry stream (row of the OTHER endpoint + one byte role/target: 5 bytes,
// coalesced), gathers the other endpoint's row (the group's own row sits in registers), recomputes
// that edge's logits — an edge is seen from both of its endpoints, 2·F·C fmas each time, nothing next
// to the gather — and forms  g = w[t]·(softmax(z) − onehot(t)).  Summed over the row's entries BEFORE
// the product with U (as the standalone dZ kernel does), separately for the entries where the row is
// the edge's src and its dst, that is S_src, S_dst:
//     dZ[r]   = (S_src·U_srcᵀ + S_dst·U_dstᵀ) / Σw          written once; rows without entries get 0
//     dU_src += Z[r]ᵀ·S_src,  dU_dst += Z[r]ᵀ·S_dst          per ROW, not per edge: the 2·F·C fp64
//                                                             accumulators are dealt over the G lanes
// No dlogits array exists, nothing is scattered, no atomics.  The entry that sees an edge from its src
// side also adds the edge's loss term and stores the logits if the caller wants them.  Σ_e w[t_e]
// depends only on the targets: the per-class edge counts come with the plan, so the gradient needs no
// second pass.
//
// Tail: one slab of fp64 partials per block, stored write-through; the LAST block to finish (device counter
// that it resets) adds the slabs in a fixed order and writes loss, dU (and dW).  One launch, bitwise
// reproducible.
//
// K > 0 ("fold"): the 1-layer model's  Z = AtXt·W  (ehf:222, F0 = K = 2) never exists: a lane gathers the
// 8-byte AtXt rows and the logits see the pre-multiplied head W·U (mathematically the same statements,
// re-associated: (x·W)·U = x·(W·U)); the per-row sums S are accumulated against AtXt instead of Z and the
// last block turns them into dU and dW.  The whole training epoch of EmbeddingGCN (cached AtXt) is this
// one kernel plus the optimizer step.
#include "common.h"

namespace tmgcn {

struct HeadLossArgs {
  const float* Z;              // [R][F]   (fold: AtXt [R][K])
  const float* Wf;             // fold: [K][F]
  const float* U;              // [2F][C]
  const int32_t* eptr;         // [R+1]
  const int4* arow;            // [n_active] (row, first entry, end entry, part): part = 0: all entries of the row; part = p > 0:
                               //   one PART of a row the plan split (long rows: hubs of the labelled edges) — its dZ share goes
                               //   to row R + p - 1 of dZ and tmgcn_head_loss_combine_f32 adds the parts up afterwards
  const int32_t* ent;          // [2E]  2*edge + role (read only when the logits are stored)
  const int32_t* other;        // [2E]  row of the other endpoint
  const uint8_t* meta;         // [2E]  role << 7 | target class (127 = ignored)
  const int64_t* class_count;  // [C]
  const float* weight;         // [C]
  const float* gscale;         // upstream gradient of the loss (device scalar) or null = 1
  float* logits;               // [E][C] or null
  float* dZ;                   // [R][F] or null
  unsigned long long* part;    // [main blocks][NP] fp64 bit patterns (stored and read write-through)
  float* loss;                 // or null
  float* dU;
  float* dW;
  int32_t* sync;
  int64_t R;
  int32_t n_active;
  int32_t main_blocks;         // blocks that walk active rows (the others only zero-fill dZ)
  // fold (K = 2) only: the optimizer step of U and W in the last block (tmgcn_head_loss_sgd_f32); sgd_on = 0 otherwise
  int32_t sgd_on;
  TmgcnSgd sgd;
  float* pU;
  float* pW;
};

template <int N, typename T>
__device__ __forceinline__ T pick(const T (&v)[N], int i) {   // v[i] for a lane-dependent i: compare chain, no scratch
  T r = v[0];
#pragma unroll
  for (int q = 1; q < N; ++q) r = (i == q) ? v[q] : r;
  return r;
}

template <int N>
__device__ __forceinline__ void load_in(const float* __restrict__ base, int64_t r, float (&v)[N]) {
  const float2* p = reinterpret_cast<const float2*>(base + r * N);
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const float2 t = p[i];
    v[2 * i] = t.x;
    v[2 * i + 1] = t.y;
  }
}

constexpr int kHeadLossMaxBlocks = 1024;   // (with the tree reduction the S2 kernel takes 30.5 - 31.0 us from 1 024 to 4 096 blocks; one last block
                                           //  walking 2 048 slabs: 81 us instead of 62, profiles/archive/r4d vs r4m)

// What a lane gathers per endpoint is the row's INPUT of NIN floats: the embedding row itself (K = 0, NIN = F)
// or, folded (K = 2), the AtXt row — then the head the logits see is H = W·U (per role), i.e.
//     logits = in_src·H_s + in_dst·H_d,      H_s = U[:F] (or W·U[:F]),  H_d = U[F:] (or W·U[F:])
// and with S_src / S_dst the per-row sums of g over the entries where the row is the edge's src / dst
//     Q_s = Σ_r in[r]ᵀ·S_src[r],  Q_d = Σ_r in[r]ᵀ·S_dst[r]        (2·NIN·C fp64 accumulators, dealt over the G lanes)
//     K = 0:  dU = [Q_s; Q_d]/Σw,   dZ[r] = (S_src·U_sᵀ + S_dst·U_dᵀ)/Σw
//     K = 2:  dU = [WᵀQ_s; WᵀQ_d]/Σw,   dW = (Q_s·U_sᵀ + Q_d·U_dᵀ)/Σw      (formed by the last block; Z, dZ never exist)
// Σ_c g[c] = 0 for every entry, so only C − 1 classes are summed; the last one is minus their sum.
// GRAD: with gradients.  Without, only the src-side entries are visited (loss, logits).  G lanes per active row.
template <int FT, int CT, bool GRAD, int K, int G>
__global__ __launch_bounds__(256) void head_loss_small_kernel(HeadLossArgs a) {
  constexpr int NIN = K ? K : FT;
  constexpr int CS = CT - 1;                          // classes whose S is summed
  constexpr int NQ = GRAD ? 2 * NIN * CT : 0;
  constexpr int NP = 1 + NQ;                          // slab: Σ w·nll | Q_s [NIN][C] | Q_d [NIN][C]
  constexpr int NPL = (NQ + G - 1) / G;               // accumulators per lane: q = gl + j·G
  constexpr int NPP = NP <= 32 ? 32 : (NP <= 64 ? 64 : 128);
  constexpr int PARTS = 256 / NPP;
  static_assert(NP <= 128, "slab wider than the finisher");
  __shared__ double red[4][NP];
  __shared__ double fin[PARTS][NPP];
  const float* __restrict__ U = a.U;
  const int gl = threadIdx.x & (G - 1);
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * 256;

  float w[CT];
  double den = 0.0;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    w[c] = a.weight[c];
    den += (double)a.class_count[c] * (double)w[c];
  }
  const double invden = 1.0 / den;                    // no labelled edge with weight: 0/0 = NaN, as torch
  const double invden_g = a.gscale ? invden * (double)a.gscale[0] : invden;
  // the head as the logits see it, per role: [NIN][CT] (uniform values: scalar registers)
  float Hs[NIN][CT], Hd[NIN][CT];
#pragma unroll
  for (int i = 0; i < NIN; ++i)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if constexpr (K == 0) {
        Hs[i][c] = U[i * CT + c];
        Hd[i][c] = U[(FT + i) * CT + c];
      } else {
        float s = 0.f, d = 0.f;
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          s = fmaf(a.Wf[i * FT + f], U[f * CT + c], s);
          d = fmaf(a.Wf[i * FT + f], U[(FT + f) * CT + c], d);
        }
        Hs[i][c] = s;
        Hd[i][c] = d;
      }
    }

  // (every uniform operand above is read before the first store of this kernel, so that it can live in scalar
  // registers: behind a store the compiler must assume it clobbered and reloads per lane)
  if constexpr (GRAD && K == 0) {                     // rows no labelled edge touches: dZ = 0
    // the sweep is left to the blocks that walk no active rows when they are the majority (the sparse regime: 186 of 1 024
    // blocks walk rows at the Bitcoin-OTC shape) — it is off the critical path of the blocks that end in the reduction
    const bool spare = (int64_t)gridDim.x >= 2 * (int64_t)a.main_blocks;
    const int64_t z_first = spare ? tid - (int64_t)a.main_blocks * 256 : tid;
    const int64_t z_step = spare ? nthreads - (int64_t)a.main_blocks * 256 : nthreads;
    for (int64_t r = z_first; r >= 0 && r < a.R; r += z_step)
      if (a.eptr[r] == a.eptr[r + 1]) {
        float2* o = reinterpret_cast<float2*>(a.dZ + r * FT);
#pragma unroll
        for (int i = 0; i < FT / 2; ++i) o[i] = make_float2(0.f, 0.f);
      }
  }
  if ((int)blockIdx.x >= a.main_blocks) return;       // block-uniform: no slab, no ticket

  double num = 0.0;
  double acc[NPL ? NPL : 1];
#pragma unroll
  for (int j = 0; j < (NPL ? NPL : 1); ++j) acc[j] = 0.0;

  const int64_t ngroups = (int64_t)a.main_blocks * 256 / G;
  int64_t i = tid / G;
  int4 cur = i < a.n_active ? a.arow[i] : make_int4(0, 0, 0, 0);
  for (; i < a.n_active; i += ngroups) {
    const int4 nxt = (i + ngroups < a.n_active) ? a.arow[i + ngroups] : make_int4(0, 0, 0, 0);   // next row's range, early
    const int64_t r = cur.x;
    float xo[NIN];
    load_in<NIN>(a.Z, r, xo);
    double S[2][CS ? CS : 1];
#pragma unroll
    for (int c = 0; c < (CS ? CS : 1); ++c) S[0][c] = S[1][c] = 0.0;
    // NB entries per lane and trip (one-lane rows are the sparse regime: two), strided over the lanes of the row: their
    // stream bytes are requested together, then their rows together — two dependent round trips per NB entries instead of
    // two per entry.  Positions past the row's end are clamped to its last entry and skipped afterwards: every load is
    // unconditional (a load under a per-lane condition becomes a branch with a full wait).  (NB CONSECUTIVE entries per
    // lane as one 16-byte load — what the layer kernels do — was measured here too: 40.2 instead of 37.5 us on S2; the
    // entries of a row are ordered by edge, so the lanes of a row gather neighbouring input rows when they take
    // neighbouring entries, and that coalescing is worth more than the saved load instructions.)
    // The per-entry terms of a trip are summed in fp32 (at most NB of them, each bounded by the class weight) and join the
    // fp64 row sums once per trip: the fp64 adds and conversions per entry were a fifth of the kernel's ≈ 100 vector
    // instructions per entry (rocprofv3 SQ counters: half of its time was VALU).
    constexpr int NB = G == 1 ? 2 : 8;               // (S2, G = 4: 8 at a time 39.6 us, 4 at a time 43.3, one at a time with G = 16 60.1, two at a time with G = 16 55.2)
    for (int pb = cur.y + gl; pb < cur.z; pb += NB * G) {
      int mb[NB], ob[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int q = pb + u * G;
        const int qc = q < cur.z ? q : cur.z - 1;
        mb[u] = a.meta[qc];
        ob[u] = a.other[qc];
      }
      float xb[NB][NIN];
#pragma unroll
      for (int u = 0; u < NB; ++u) load_in<NIN>(a.Z, ob[u], xb[u]);
      float Sp[2][CS ? CS : 1], nump = 0.f;
#pragma unroll
      for (int c = 0; c < (CS ? CS : 1); ++c) Sp[0][c] = Sp[1][c] = 0.f;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
      const int p = pb + u * G;
      if (p >= cur.z) continue;
      const int m = mb[u];
      const bool role = m & 0x80;
      if (!GRAD && role) continue;
      const int t = m & 0x7f;
      float xt[NIN];
#pragma unroll
      for (int q = 0; q < NIN; ++q) xt[q] = xb[u][q];
      // logits: input index ascending, src then dst (K = 0: the fmaf chain of edge_head_fwd_small, from either endpoint)
      float lg[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) lg[c] = 0.f;
#pragma unroll
      for (int q = 0; q < NIN; ++q) {
        const float sv = role ? xt[q] : xo[q], dv = role ? xo[q] : xt[q];
#pragma unroll
        for (int c = 0; c < CT; ++c) lg[c] = fmaf(dv, Hd[q][c], fmaf(sv, Hs[q][c], lg[c]));
      }
      float mx = lg[0];
#pragma unroll
      for (int c = 1; c < CT; ++c) mx = fmaxf(mx, lg[c]);
      float ex[CT], s = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        ex[c] = __expf(lg[c] - mx);                   // arguments <= 0: the hardware exp2 is within 2 ulp there
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
      if constexpr (GRAD) {
        const float inv = __builtin_amdgcn_rcpf(s);            // 1 ulp
#pragma unroll
        for (int c = 0; c < CS; ++c) {
          const float g = valid ? wt * (ex[c] * inv - (c == t ? 1.f : 0.f)) : 0.f;   // an ignored entry adds an exact 0, whatever its logits
          Sp[0][c] += role ? 0.f : g;
          Sp[1][c] += role ? g : 0.f;
        }
      }
      if (!role) {
        // s is in [1, C]: the hardware log2 needs no denormal handling there (1 ulp)
        nump += valid ? wt * ((mx - zt_t) + __builtin_amdgcn_logf(s) * 0.693147180559945309f) : 0.f;
        if (a.logits) {
          float* o = a.logits + (int64_t)(a.ent[p] >> 1) * CT;
          if constexpr (CT == 2) {
            *reinterpret_cast<float2*>(o) = make_float2(lg[0], lg[1]);
          } else if constexpr (CT == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(lg[0], lg[1], lg[2], lg[3]);
          } else {
#pragma unroll
            for (int c = 0; c < CT; ++c) o[c] = lg[c];
          }
        }
      }
      }
      num += (double)nump;
      if constexpr (GRAD) {
#pragma unroll
        for (int c = 0; c < CS; ++c) {
          S[0][c] += (double)Sp[0][c];
          S[1][c] += (double)Sp[1][c];
        }
      }
    }
    if constexpr (GRAD) {
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1) {
#pragma unroll
        for (int c = 0; c < CS; ++c) {
          S[0][c] += __shfl_xor(S[0][c], o);
          S[1][c] += __shfl_xor(S[1][c], o);
        }
      }
      double Sf[2 * CT];                              // [role][c], all classes
      {
        double l0 = 0.0, l1 = 0.0;
#pragma unroll
        for (int c = 0; c < CS; ++c) {
          Sf[c] = S[0][c];
          Sf[CT + c] = S[1][c];
          l0 -= S[0][c];
          l1 -= S[1][c];
        }
        Sf[CS] = l0;
        Sf[CT + CS] = l1;
      }
      if constexpr (K == 0) {
        float sf[2 * CT];
#pragma unroll
        for (int c = 0; c < 2 * CT; ++c) sf[c] = (float)(Sf[c] * invden_g);
        // (a part of a split row stores its share of dZ[r] — the expression is linear in S — to its own scratch row)
        float2* o = reinterpret_cast<float2*>(a.dZ + (cur.w ? a.R + cur.w - 1 : r) * FT);
        for (int q = gl; q < FT / 2; q += G) {        // lane q of the group stores features 2q, 2q+1
          float v0 = 0.f, v1 = 0.f;
#pragma unroll
          for (int h = 0; h < FT / 2; ++h)
            if (q == h) {
#pragma unroll
              for (int c = 0; c < CT; ++c) {
                v0 = fmaf(sf[CT + c], Hd[2 * h][c], fmaf(sf[c], Hs[2 * h][c], v0));
                v1 = fmaf(sf[CT + c], Hd[2 * h + 1][c], fmaf(sf[c], Hs[2 * h + 1][c], v1));
              }
            }
          o[q] = make_float2(v0, v1);
        }
      }
      // lane gl owns the accumulators q = gl + j·G of Q[role][in][c]:  += in[r][in]·S[role][c]
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int qi = gl + j * G;
        if (qi < NQ) {
          const int role = qi / (NIN * CT), rem = qi - role * NIN * CT, in = rem / CT, c = rem - in * CT;
          acc[j] = fma((double)pick<NIN>(xo, in), pick<2 * CT>(Sf, role * CT + c), acc[j]);
        }
      }
    }
    cur = nxt;
  }

  // block totals: lanes with the same gl hold the same accumulators — butterfly over the lane bits above G,
  // the four waves in order through LDS; the loss term over all lanes.  Fixed order: reproducible.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) num += __shfl_xor(num, o);
  if (lane == 0) red[wave][0] = num;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    double v = acc[j];
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) v += __shfl_xor(v, o);
    const int qi = gl + j * G;
    if (lane < G && qi < NQ) red[wave][1 + qi] = v;
  }
  __syncthreads();
  // hand-off to the last block (cdna_hip_programming.md §6 Guideline 16, recipe R1): the slab is stored
  // WRITE-THROUGH (8-byte relaxed agent-scope atomic stores = sc1), every storing wave drains its stores, the
  // block meets, ONE lane takes a ticket; the block that draws the last ticket reads every slab with sc1 loads
  // (relaxed agent-scope atomic loads).  No release / acquire fence: nothing else is handed over, and the XCDs'
  // L2s — not coherent with each other — are bypassed both ways.
  if (threadIdx.x < NP) {
    const double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    __hip_atomic_store(a.part + (int64_t)blockIdx.x * NP + threadIdx.x, (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the slabs are added as a tree over the two ticket levels (common.h: slab_tree_finish_f64) — one last block walking the
  // 1 024 slabs of a Reddit-LP-sized launch with 8 threads per output and 8 loads in flight took 16 round trips
  __shared__ double totals[NP];
  if (!slab_tree_finish_f64<NP>(a.part, a.main_blocks, a.sync, totals)) return;
  if (threadIdx.x < NP) {
    const double total = totals[threadIdx.x];
    fin[0][threadIdx.x] = total;                      // Q totals, for the folded outputs below
    if (threadIdx.x == 0) {
      if (a.loss) a.loss[0] = (float)(total * invden);      // (the launch's counters are zero again: slab_tree_finish_f64)
    }
    if constexpr (GRAD && K == 0) {
      if (threadIdx.x >= 1) a.dU[threadIdx.x - 1] = (float)(total * invden_g);      // Q = [Q_s; Q_d] is dU's layout
    }
  }
  if constexpr (GRAD && K != 0) {
    __syncthreads();
    const double* Q = &fin[0][1];                     // [role][k][c]
    const int t = threadIdx.x;
    if (t < 2 * FT * CT) {                            // dU[role·F + f][c] = Σ_k W[k][f]·Q[role][k][c]
      const int role = t / (FT * CT), rem = t - role * FT * CT, f = rem / CT, c = rem - f * CT;
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) v = fma((double)a.Wf[k * FT + f], Q[(role * K + k) * CT + c], v);
      a.dU[t] = (float)(v * invden_g);
    } else if (t < 2 * FT * CT + K * FT) {            // dW[k][f] = Σ_c Q_s[k][c]·U[f][c] + Q_d[k][c]·U[F+f][c]
      const int b = t - 2 * FT * CT, k = b / FT, f = b - k * FT;
      double v = 0.0;
#pragma unroll
      for (int c = 0; c < CT; ++c)
        v = fma(Q[(K + k) * CT + c], (double)U[(FT + f) * CT + c], fma(Q[k * CT + c], (double)U[f * CT + c], v));
      a.dW[b] = (float)(v * invden_g);
    }
    if (a.sgd_on) {
      // the optimizer step (tmgcn_sgd_step's arithmetic), each thread on the element whose gradient it just formed; the
      // block meets first: the gradients above read U and W
      __syncthreads();
      float* p = nullptr;
      float* buf = nullptr;
      int i = 0;
      if (t < 2 * FT * CT) p = a.pU, buf = a.sgd.buf_U, i = t;
      else if (t < 2 * FT * CT + K * FT) p = a.pW, buf = a.sgd.buf_W, i = t - 2 * FT * CT;
      if (p) {
        const float w0 = p[i];
        float g = (t < 2 * FT * CT) ? a.dU[i] : a.dW[i];
        if (a.sgd.maximize) g = -g;
        if (a.sgd.weight_decay != 0.f) g = fmaf(a.sgd.weight_decay, w0, g);
        if (a.sgd.momentum != 0.f) {
          float b;
          if (a.sgd.first_step) b = g;
          else b = __fmul_rn(a.sgd.momentum, buf[i]) + (1.f - a.sgd.dampening) * g;
          buf[i] = b;
          g = a.sgd.nesterov ? fmaf(a.sgd.momentum, b, g) : b;
        }
        p[i] = fmaf(-a.sgd.lr, g, w0);
      }
    }
  }
}

static int head_loss_lanes(int64_t E, int64_t n_active) {   // lanes per active row: chains of about two entries
  const int64_t avg = n_active > 0 ? (2 * E + n_active - 1) / n_active : 0;
  return avg <= 4 ? 1 : (avg <= 32 ? 4 : 16);   // a lane walks its entries NB at a time
}

template <int FT, int CT, bool GRAD, int K>
static void head_loss_launch_g(HeadLossArgs a, int G, int64_t R, hipStream_t st) {
  int64_t main_blocks = ((int64_t)a.n_active * G + 255) / 256;
  if (main_blocks > kHeadLossMaxBlocks) main_blocks = kHeadLossMaxBlocks;
  if (main_blocks < 1) main_blocks = 1;
  int64_t blocks = main_blocks;
  if (GRAD && K == 0) {                               // the zero-fill sweep may use more blocks than there are active rows
    const int64_t z = (R + 255) / 256;
    blocks = z > blocks ? z : blocks;
    if (blocks > kHeadLossMaxBlocks) blocks = kHeadLossMaxBlocks;
  }
  a.main_blocks = (int32_t)main_blocks;
  const dim3 grid((unsigned)blocks), blk(256);
  switch (G) {
    case 1: hipLaunchKernelGGL((head_loss_small_kernel<FT, CT, GRAD, K, 1>), grid, blk, 0, st, a); break;
    case 4: hipLaunchKernelGGL((head_loss_small_kernel<FT, CT, GRAD, K, 4>), grid, blk, 0, st, a); break;
    default: hipLaunchKernelGGL((head_loss_small_kernel<FT, CT, GRAD, K, 16>), grid, blk, 0, st, a);
  }
}

template <int FT, int CT>
static void head_loss_launch(const HeadLossArgs& a, bool grad, int K, int G, int64_t R, hipStream_t st) {
  if (K == 0) {
    if (grad) head_loss_launch_g<FT, CT, true, 0>(a, G, R, st);
    else head_loss_launch_g<FT, CT, false, 0>(a, G, R, st);
  } else {
    if (grad) head_loss_launch_g<FT, CT, true, 2>(a, G, R, st);
    else head_loss_launch_g<FT, CT, false, 2>(a, G, R, st);
  }
}

template <int FT>
static void head_loss_launch_c(const HeadLossArgs& a, int C, bool grad, int K, int G, int64_t R, hipStream_t st) {
  switch (C) {
    case 2: head_loss_launch<FT, 2>(a, grad, K, G, R, st); break;
    case 3: head_loss_launch<FT, 3>(a, grad, K, G, R, st); break;
    default: head_loss_launch<FT, 4>(a, grad, K, G, R, st);
  }
}

// dZ of the rows the plan split: dZ[row] = Σ_parts dZ[R + first + k], k ascending (fixed order).  One thread per (row, feature).
__global__ __launch_bounds__(256) void head_loss_combine_kernel(const int4* __restrict__ srow, int32_t n_split, float* __restrict__ dZ,
                                                                 int64_t R, int32_t F) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)n_split * F) return;
  const int4 s = srow[i / F];                         // (row, first part, number of parts, -)
  const int f = (int)(i % F);
  float v = 0.f;
  for (int k = 0; k < s.z; ++k) v += dZ[(R + s.y + k) * F + f];
  dZ[(int64_t)s.x * F + f] = v;
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
  return (F >= 2 && F <= 8 && F % 2 == 0 && C >= 2 && C <= 4 && (K == 0 || K == 2)) ? 1 : 0;
}

extern "C" int64_t tmgcn_head_loss_workspace_bytes(int32_t F, int32_t C, int32_t K) {
  if (!tmgcn_head_loss_supported(F, C, K)) return 0;
  return (int64_t)(kHeadLossMaxBlocks + kSyncGroups) * (1 + 2 * (K ? K : F) * C) * (int64_t)sizeof(double);   // block slabs + group slabs
}

static thread_local const TmgcnSgd* g_head_loss_sgd = nullptr;

extern "C" int tmgcn_head_loss_f32(const float* Z, const float* W_fold, int32_t K, const float* U,
                                    const int32_t* eptr, const int32_t* arow, int64_t n_active, const int32_t* ent,
                                    const int32_t* other, const uint8_t* meta, const int64_t* class_count,
                                    const float* weight, const float* grad_scale, int64_t R, int64_t E, int32_t F,
                                    int32_t C, float* logits, float* loss, float* dZ, float* dU, float* dW,
                                    void* workspace, int64_t workspace_bytes, int32_t* sync, void* stream) {
  TMGCN_REQUIRE(tmgcn_head_loss_supported(F, C, K), "head_loss: unsupported widths F=%d C=%d K=%d (even F <= 8, 2 <= C <= 4, K in {0, 2})",
                F, C, K);
  TMGCN_REQUIRE(R > 0 && E > 0 && R < (int64_t)0x7fffffff && 2 * E < (int64_t)0x7fffffff && n_active > 0 && n_active <= R + 2 * E &&
                    n_active < (int64_t)0x7fffffff,
                "head_loss: need 0 < R, 2E < 2^31 and 0 < n_active <= R + 2E, n_active < 2^31 (got R=%lld E=%lld n_active=%lld)", (long long)R,
                (long long)E, (long long)n_active);
  TMGCN_REQUIRE(Z && U && eptr && arow && other && meta && class_count && weight && workspace, "head_loss: null pointer");
  if (!sync) sync = acquire_sync_word((hipStream_t)stream);
  TMGCN_REQUIRE(sync, "head_loss: no hand-off block: %s", pool_error());
  TMGCN_REQUIRE(!logits || ent, "head_loss: the logits need the entry -> edge index (ent)");
  TMGCN_REQUIRE((K == 0) == (W_fold == nullptr), "head_loss: W_fold must be given exactly when K > 0");
  const bool grad = dU != nullptr;
  TMGCN_REQUIRE(grad || loss, "head_loss: nothing to compute (neither loss nor gradients asked for)");
  TMGCN_REQUIRE(!grad || (K ? dW != nullptr : dZ != nullptr), "head_loss: gradients asked for (dU) but no %s", K ? "dW" : "dZ");
  TMGCN_REQUIRE(reinterpret_cast<uintptr_t>(Z) % 8 == 0 && (!dZ || reinterpret_cast<uintptr_t>(dZ) % 8 == 0) &&
                    (!logits || reinterpret_cast<uintptr_t>(logits) % 16 == 0) && reinterpret_cast<uintptr_t>(arow) % 16 == 0,
                "head_loss: Z / dZ must be 8-byte aligned, logits and arow 16-byte aligned");
  if (workspace_bytes < tmgcn_head_loss_workspace_bytes(F, C, K)) {
    set_error("head_loss: workspace %lld B < required %lld B", (long long)workspace_bytes,
              (long long)tmgcn_head_loss_workspace_bytes(F, C, K));
    return TMGCN_ERR_WORKSPACE;
  }
  HeadLossArgs a{Z, W_fold, U, eptr, reinterpret_cast<const int4*>(arow), ent, other, meta, class_count, weight, grad_scale,
                 logits, dZ, (unsigned long long*)workspace, loss, dU, dW, sync, R, (int32_t)n_active, 0, 0, TmgcnSgd{}, nullptr, nullptr};
  if (g_head_loss_sgd) {                              // set by tmgcn_head_loss_sgd_f32 for this one call (same thread)
    a.sgd_on = 1, a.sgd = *g_head_loss_sgd, a.pU = const_cast<float*>(U), a.pW = const_cast<float*>(W_fold);
  }
  const int G = head_loss_lanes(E, n_active);
  hipStream_t st = (hipStream_t)stream;
  switch (F) {
    case 2: head_loss_launch_c<2>(a, C, grad, K, G, R, st); break;
    case 4: head_loss_launch_c<4>(a, C, grad, K, G, R, st); break;
    case 6: head_loss_launch_c<6>(a, C, grad, K, G, R, st); break;
    default: head_loss_launch_c<8>(a, C, grad, K, G, R, st);
  }
  return check_launch("head_loss");
}

extern "C" int tmgcn_head_loss_lanes(int64_t E, int64_t n_active) { return head_loss_lanes(E, n_active); }

extern "C" int tmgcn_head_loss_combine_f32(const int32_t* srow, int32_t n_split, float* dZ, int64_t R, int32_t F, void* stream) {
  TMGCN_REQUIRE(n_split >= 0 && R > 0 && F > 0, "head_loss_combine: bad sizes n_split=%d R=%lld F=%d", n_split, (long long)R, F);
  if (n_split == 0) return TMGCN_OK;
  TMGCN_REQUIRE(srow && dZ && reinterpret_cast<uintptr_t>(srow) % 16 == 0, "head_loss_combine: null pointer or srow not 16-byte aligned");
  const int64_t n = (int64_t)n_split * F;
  hipLaunchKernelGGL(head_loss_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const int4*>(srow), n_split, dZ, R, F);
  return check_launch("head_loss_combine");
}

extern "C" int tmgcn_head_loss_sgd_f32(const float* Z, float* W_fold, int32_t K, float* U, const int32_t* eptr, const int32_t* arow,
                                        int64_t n_active, const int32_t* other, const uint8_t* meta, const int64_t* class_count,
                                        const float* weight, int64_t R, int64_t E, int32_t F, int32_t C, float* loss, float* dU,
                                        float* dW, const TmgcnSgd* sgd, void* workspace, int64_t workspace_bytes, int32_t* sync,
                                        void* stream) {
  TMGCN_REQUIRE(K == 2 && W_fold && sgd && dU && dW, "head_loss_sgd: the folded form (K = 2) with dU, dW and the optimizer settings");
  TMGCN_REQUIRE(sgd->lr >= 0.f && sgd->momentum >= 0.f && sgd->weight_decay >= 0.f, "head_loss_sgd: negative hyper-parameter");
  TMGCN_REQUIRE(sgd->momentum == 0.f || (sgd->buf_U && sgd->buf_W), "head_loss_sgd: momentum needs both buffers");
  g_head_loss_sgd = sgd;
  const int rc = tmgcn_head_loss_f32(Z, W_fold, K, U, eptr, arow, n_active, nullptr, other, meta, class_count, weight, nullptr, R, E, F,
                                     C, nullptr, loss, nullptr, dU, dW, workspace, workspace_bytes, sync, stream);
  g_head_loss_sgd = nullptr;
  return rc;
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

// Shared helpers for the gfx950 kernels of the TM-GCN layer (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>
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

// Work counters for dynamically scheduled persistent kernels (pools.hip owns them).
// Persistent blocks draw their next tile from a device counter instead of a static
// blockIdx-strided assignment: when some blocks are not resident from the start (another
// kernel — e.g. RCCL's all-to-all on a side stream — holds CUs), the late blocks simply draw
// fewer tiles instead of serialising a full share behind the others; skewed row lengths balance
// the same way.  acquire_tile_counter() returns a zeroed counter (memset enqueued on `stream`) that no launch on
// another stream, and no recorded launch of another hipGraph, shares (pools.hip); nullptr — with the reason in
// pool_error() — when it cannot promise that.
const char* pool_error();
unsigned int* acquire_tile_counters(hipStream_t stream, int n);  // n <= 64 consecutive zeroed counters
inline unsigned int* acquire_tile_counter(hipStream_t stream) { return acquire_tile_counters(stream, 1); }

// A hand-off block (kSyncInts int32, last_block_ticket) for a kernel whose last block finishes a reduction
// (pools.hip): zero when the launch starts, left zero by the launch — no memset node per call.  One block per stream
// for eager launches, one for good per recorded launch; nullptr (reason in pool_error()) when none can be given.
int32_t* acquire_sync_word(hipStream_t stream);

// Called by EVERY thread of a block after its slab stores (write-through: 4- / 8-byte relaxed agent-scope atomic
// stores): drains the stores, meets, takes ONE ticket; true in every thread of the block that drew the last one.
// That block then reads the slabs with relaxed agent-scope atomic loads (cdna_hip_programming.md §6 Guideline 16,
// recipe R1: no release / acquire fence when every handed-off byte is stored and loaded write-through).
// Tickets are drawn in TWO levels: agent-scope atomics on one address are served one after the other by the memory side
// (the XCDs' L2s are not coherent), ≈ 16 ns each — a kernel whose 1 024 blocks finish together queued for 16 us on a
// single counter (narrow dW: 13.1 / 17.2 / 26.8 us with 279 / 557 / 1 114 blocks, the same bytes).  So block b draws
// from counter 1 + b mod 16 of the launch's hand-off block (kSyncInts int32, the counters 64 bytes apart), the last of
// each of those sixteen groups draws from counter 0, and the last there is the last of all.  The group counters are
// left zero by their last drawer, counter 0 by the caller (`*sync = 0` in the finishing block).
constexpr int kSyncGroups = 16, kSyncStride = 16;
constexpr int kSyncInts = (1 + kSyncGroups) * kSyncStride;
static_assert(kSyncInts == TMGCN_SYNC_INTS, "include/tmgcn.h states the size of a hand-off block");
__device__ __forceinline__ bool last_block_ticket(int32_t* sync, int n_blocks, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int g = blockIdx.x % kSyncGroups;
    const int members = (n_blocks - g + kSyncGroups - 1) / kSyncGroups;     // blocks b < n_blocks with b mod 16 == g
    int32_t* mine = sync + (1 + g) * kSyncStride;
    bool last = false;
    if (__hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
      __hip_atomic_store(mine, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int groups = n_blocks < kSyncGroups ? n_blocks : kSyncGroups;
      last = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
    }
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// Giant-row pre-pass (spmm.hip; used by the plain and the fused SpMM launchers): validates the plan, launches one block per
// chunk that writes the chunk's partial sum of Â·X to `ws`.  Returns TMGCN_OK without a launch when n_giant == 0.
int launch_giant_partial(const char* who, const int64_t* rowptr, const int32_t* col, const float* val, const float* X, int32_t N,
                         int32_t F, const int64_t* giant_rows, const int32_t* giant_chunks, int32_t n_giant, int32_t n_giant_chunks,
                         float* ws, int64_t ws_bytes, hipStream_t st);

#define TMGCN_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ::tmgcn::set_error(__VA_ARGS__);      \
      return TMGCN_ERR_INVALID;             \
    }                                       \
  } while (0)

constexpr int kWave = 64;  // gfx950 wavefront

// Grid size of a persistent kernel: CUs x resident blocks per CU (occupancy API, capped at 4:
// MI355X_MICROARCH.md warns the API can over-report by one for SGPR-heavy kernels; a persistent
// grid that is not fully resident runs its tail blocks serially).  Cached per kernel.

// CU count of a device, queried once per device and host thread (hipGetDeviceProperties fills a
// multi-KB struct: too slow for the launch path of 0.2 ms epochs).
inline int device_cu_count(int dev) {
  thread_local int cus[16] = {0};
  if (dev >= 0 && dev < 16 && cus[dev]) return cus[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0 && dev < 16) cus[dev] = n;
  return n;
}

template <typename K>
inline int persistent_grid(K kernel, int block_threads, size_t dyn_smem = 0, int max_per_cu = 4) {
  // K is the function-pointer TYPE, which every kernel with the same argument list shares, so the
  // cache is a small per-thread table keyed on (kernel address, device, block size): alternating
  // between instantiations (fp32 / bf16 weights, layers of different widths) hits it every time.
  // No shared mutable state between host threads.  Dynamic LDS changes residency: not cached.
  struct Entry { const void* k; int dev, threads, blocks_per_cu; };
  constexpr int kSlots = 16;
  thread_local Entry table[kSlots] = {};
  thread_local int next = 0;
  const void* key = reinterpret_cast<const void*>(kernel);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const int cus = device_cu_count(dev);
  if (dyn_smem == 0)
    for (int i = 0; i < kSlots; ++i)
      if (table[i].k == key && table[i].dev == dev && table[i].threads == block_threads)
        return cus * table[i].blocks_per_cu;
  int per_cu = 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, dyn_smem) != hipSuccess ||
      per_cu < 1)
    per_cu = 1;
  if (per_cu > max_per_cu) per_cu = max_per_cu;
  (void)hipGetLastError();
  if (dyn_smem == 0) {
    table[next] = Entry{key, dev, block_threads, per_cu};
    next = (next + 1) % kSlots;
  }
  return cus * per_cu;
}

// Persistent grid minus `reserve` block slots (a per-call argument of the C-ABI): in the multi-GPU
// path RCCL's kernels run on a side stream and can only become resident if the persistent compute
// kernel does not hold every block slot of every CU.
template <typename K>
inline int persistent_grid_reserved(K kernel, int block_threads, int reserve, size_t dyn_smem = 0) {
  int g = persistent_grid(kernel, block_threads, dyn_smem) - reserve;
  return g < 64 ? 64 : g;
}

// 4x4 transpose across a lane quad: on entry register k of lane j (j = lane & 3) holds M[k][j], on
// exit it holds M[j][k].  The MFMA accumulator layout has the output COLUMN on the lane and four
// consecutive ROWS in consecutive registers, so a plain epilogue stores one dword per lane per row
// (two 128-B segments per wave-instruction) and is bound by store ISSUE, not by bandwidth
// (MI355X_MICROARCH.md, "epilogue store tail"; cdna_hip_programming.md T21).  After this transpose
// lane j of a quad owns row j and four consecutive columns: one 16-byte store per lane, eight full
// 128-B lines per wave-instruction, a quarter of the store instructions.  Two butterfly stages of
// quad-permute DPP moves (no LDS): 12 VALU instructions per 4x4 block.
__device__ __forceinline__ void quad_transpose4(float (&v)[4], int j) {
  const bool odd = j & 1, hi = j & 2;
  {  // stage 1: lanes j ^ 1, register pairs (0,1) and (2,3)
    const float s0 = odd ? v[0] : v[1], s1 = odd ? v[2] : v[3];
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
    if (odd) {
      v[0] = r0;
      v[2] = r1;
    } else {
      v[1] = r0;
      v[3] = r1;
    }
  }
  {  // stage 2: lanes j ^ 2, register pairs (0,2) and (1,3)
    const float s0 = hi ? v[0] : v[2], s1 = hi ? v[1] : v[3];
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0x4E, 0xF, 0xF, true));
    if (hi) {
      v[0] = r0;
      v[1] = r1;
    } else {
      v[2] = r0;
      v[3] = r1;
    }
  }
}

// Σ over the 64 lanes of a wave, the same value in every lane, without the LDS crossbar: four DPP steps inside each row of
// 16 lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then the four rows' sums read as scalars and added
// in row order.  A fixed order (reproducible); 11 VALU operations where a __shfl_xor butterfly is six ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_lane_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += dpp_lane_f32<0xB1>(v);
  v += dpp_lane_f32<0x4E>(v);
  v += dpp_lane_f32<0x141>(v);
  v += dpp_lane_f32<0x140>(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return ((r0 + r1) + r2) + r3;
}

template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double x) {          // x + (x of the lane the DPP control names): two 32-bit moves
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
  return x + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {        // the same steps on the two halves of an fp64
  auto step = [](double x, auto ctrl) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, decltype(ctrl)::value, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), decltype(ctrl)::value, 0xF, 0xF, true);
    return x + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
  };
  v = step(v, std::integral_constant<int, 0xB1>{});
  v = step(v, std::integral_constant<int, 0x4E>{});
  v = step(v, std::integral_constant<int, 0x141>{});
  v = step(v, std::integral_constant<int, 0x140>{});
  auto lane = [&](int l) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
  };
  return ((lane(0) + lane(16)) + lane(32)) + lane(48);
}

// A thread's share of the last block's slab reduction: Σ slab[c][o] over the slabs c = first, first + stride, … below n,
// read with sc1 loads (see last_block_ticket), DEPTH of them in flight — issued together and masked, not branched, past
// the end — and added in slab order (reproducible).  The loads come from memory, about a microsecond a round trip: with
// 8 in flight the last block of the fused backward spent 6 us adding 1 024 slabs; 32 bring that to two round trips.
#ifndef TMGCN_FIN_DEPTH
#define TMGCN_FIN_DEPTH 8
#endif
constexpr int kFinisherDepth = TMGCN_FIN_DEPTH;

template <int DEPTH>
__device__ __forceinline__ double slab_sum_f32(const unsigned* slabs, int n, int first, int stride, int width, int o) {
  double s = 0.0;
  for (int c = first; c < n; c += DEPTH * stride) {
    float v[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      const int cc = c + q * stride;
      v[q] = __uint_as_float(__hip_atomic_load(slabs + (int64_t)(cc < n ? cc : n - 1) * width + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) s += c + q * stride < n ? (double)v[q] : 0.0;
  }
  return s;
}

// The hand-off of a slab reduction as a TREE over the two ticket levels of last_block_ticket: the block that draws the last
// ticket of its group (blocks b ≡ g mod 16) adds the slabs of that group and stores a GROUP slab; the block that draws the
// last of the sixteen top-level tickets adds the group slabs.  Every reading is one or two round trips whatever the grid
// (one last block walking 1 024 slabs with 8 loads in flight took 7 round trips, 2 282 slabs 14: why the backward grids
// were capped at 1 024 blocks), all sums in a fixed order.  Called by every thread of every block after the block's own
// slab (NO floats, write-through) is stored at part[blockIdx.x]; `part` holds n_blocks + kSyncGroups slabs.  True in the
// one block that ends up with the totals, in total[0 .. NO) (LDS); that block has also reset the launch's counters.
// `n_slabs` (default: one per block) — the slabs may also belong to units of work the blocks of a group share (slab s to
// group s mod 16: the entry-major layer backward stores one per ROW BLOCK, whichever block of the group worked on it).
// slab_tree_finish_in: the same with the caller's LDS for the partial sums (`fin`: 256 / NO · NO doubles — a kernel whose
// tile buffers are free by then stays under the LDS of one more resident block).
// `ranges`: group g's slabs are the consecutive ones [n_slabs·g / groups, n_slabs·(g + 1) / groups) instead of those ≡ g mod 16
// (the entry-major layer backward gives every group — the thread blocks of one XCD — a run of neighbouring row blocks).
template <int NO>
__device__ __forceinline__ bool slab_tree_finish_in(unsigned* part, int n_blocks, int32_t* sync, double* total /* LDS [NO] */, int n_slabs,
                                                    double* fin, bool ranges = false) {
  constexpr int SUBS = 256 / NO;
  double (*tree_fin)[NO] = reinterpret_cast<double (*)[NO]>(fin);
  __shared__ int tree_flag;
  if (n_slabs < 0) n_slabs = n_blocks;
  const int g = blockIdx.x % kSyncGroups;
  const int members = (n_blocks - g + kSyncGroups - 1) / kSyncGroups;
  const int groups = n_blocks < kSyncGroups ? n_blocks : kSyncGroups;
  const int64_t range_lo = (int64_t)n_slabs * g / groups;
  const int slabs = ranges ? (int)((int64_t)n_slabs * (g + 1) / groups - range_lo) : (n_slabs - g + kSyncGroups - 1) / kSyncGroups;   // of this group
  int32_t* mine = sync + (1 + g) * kSyncStride;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool last = __hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1;
    if (last) __hip_atomic_store(mine, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tree_flag = last;
  }
  __syncthreads();
  if (!tree_flag) return false;
  // the group's slabs g, g + 16, …: SUBS threads per output, then the SUBS partial sums in order
  const int sub = threadIdx.x / NO, o = threadIdx.x - sub * NO;
  if (sub < SUBS)
    tree_fin[sub][o] = ranges ? slab_sum_f32<8>(part + range_lo * NO, slabs, sub, SUBS, NO, o)
                              : slab_sum_f32<8>(part + (int64_t)g * NO, slabs, sub, SUBS, NO * kSyncGroups, o);
  __syncthreads();
  if (threadIdx.x < NO) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < SUBS; ++q) t += tree_fin[q][threadIdx.x];
    __hip_atomic_store(part + (int64_t)(n_slabs + g) * NO + threadIdx.x, __float_as_uint((float)t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) tree_flag = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
  __syncthreads();
  if (!tree_flag) return false;
  if (threadIdx.x == 0) __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < NO) total[threadIdx.x] = slab_sum_f32<kSyncGroups>(part + (int64_t)n_slabs * NO, groups, 0, 1, NO, threadIdx.x);
  __syncthreads();
  return true;
}

template <int NO>
__device__ __forceinline__ bool slab_tree_finish(unsigned* part, int n_blocks, int32_t* sync, double* total /* LDS [NO] */, int n_slabs = -1) {
  __shared__ double tree_fin_own[256 / NO][NO];
  return slab_tree_finish_in<NO>(part, n_blocks, sync, total, n_slabs, &tree_fin_own[0][0]);
}

// The same tree for slabs of fp64 bit patterns (8-byte write-through stores / loads): the head + loss kernel.
template <int DEPTH>
__device__ __forceinline__ double slab_sum_f64(const unsigned long long* slabs, int n, int first, int stride, int width, int o) {
  double s = 0.0;
  for (int c = first; c < n; c += DEPTH * stride) {
    double v[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      const int cc = c + q * stride;
      v[q] = __longlong_as_double((long long)__hip_atomic_load(slabs + (int64_t)(cc < n ? cc : n - 1) * width + o, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT));
    }
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) s += c + q * stride < n ? v[q] : 0.0;
  }
  return s;
}

template <int NO>
__device__ __forceinline__ bool slab_tree_finish_f64(unsigned long long* part, int n_blocks, int32_t* sync, double* total /* LDS [NO] */) {
  constexpr int SUBS = 256 / NO;
  __shared__ double tree_fin64[SUBS][NO];
  __shared__ int tree_flag64;
  const int g = blockIdx.x % kSyncGroups;
  const int members = (n_blocks - g + kSyncGroups - 1) / kSyncGroups;
  const int groups = n_blocks < kSyncGroups ? n_blocks : kSyncGroups;
  int32_t* mine = sync + (1 + g) * kSyncStride;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool last = __hip_atomic_fetch_add(mine, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1;
    if (last) __hip_atomic_store(mine, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tree_flag64 = last;
  }
  __syncthreads();
  if (!tree_flag64) return false;
  const int sub = threadIdx.x / NO, o = threadIdx.x - sub * NO;
  if (sub < SUBS) tree_fin64[sub][o] = slab_sum_f64<8>(part + (int64_t)g * NO, members, sub, SUBS, NO * kSyncGroups, o);
  __syncthreads();
  if (threadIdx.x < NO) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < SUBS; ++q) t += tree_fin64[q][threadIdx.x];
    __hip_atomic_store(part + (int64_t)(n_blocks + g) * NO + threadIdx.x, (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) tree_flag64 = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
  __syncthreads();
  if (!tree_flag64) return false;
  if (threadIdx.x == 0) __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < NO) total[threadIdx.x] = slab_sum_f64<kSyncGroups>(part + (int64_t)n_blocks * NO, groups, 0, 1, NO, threadIdx.x);
  __syncthreads();
  return true;
}

// e^x on the negative side of SELU and of its derivative (x <= 0): one multiply and the hardware exp2 instead of expf's
// range reduction and overflow handling (2 VALU operations for about 12 — the fused layer kernels evaluate it per gathered
// non-zero and were bound by exactly that: 1 180 vector instructions per wave of the Bitcoin-OTC-shaped forward, rocprofv3 SQ
// counters, round 4).  Relative error <= 2 ulp + |x|·6e-8 of a value <= 1: below 1.2e-7 absolute, against torch's SELU.
// Every activation path of the library (act_apply / act_grad, ActApply / ActGrad) shares it, so fused and unfused routes
// keep producing the same bits.
__device__ __forceinline__ float exp_nonpos(float x) { return __expf(fminf(x, 0.f)); }

__device__ __forceinline__ float act_apply(float x, int act) {
  // torch.nn.ReLU / LeakyReLU(0.01) / SELU constants (ehf:284-289)
  switch (act) {
    case TMGCN_ACT_RELU: return x > 0.f ? x : 0.f;
    case TMGCN_ACT_LEAKY: return x > 0.f ? x : 0.01f * x;
    case TMGCN_ACT_SELU: {
      const float scale = 1.0507009873554804934193349852946f;
      const float alpha = 1.6732632423543772848170429916717f;
      return x > 0.f ? scale * x : scale * alpha * (exp_nonpos(x) - 1.f);
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
      return x > 0.f ? scale : scale * alpha * exp_nonpos(x);   // exp evaluated unconditionally: a select, not a branch
    }
    default: return 1.f;
  }
}

// act_apply() with the activation decoded ONCE (uniform registers) instead of a switch per element: the same values, bit
// for bit — x > 0 ? pos·x : neg·(use_exp ? exp(x) − 1 : x), relu's negative side an exact +0 — as selects, so an
// unrolled loop over elements carries no scalar branch chain and no exec-mask region per element (the per-element
// switch cost the fused layer kernel 636 scalar and ≈ 400 vector instructions per row; rocprofv3 SQ counters, round 4).
struct ActApply {
  float pos, neg;
  bool use_exp, zero_neg;
  __device__ __forceinline__ explicit ActApply(int act) {
    const float scale = 1.0507009873554804934193349852946f;
    const float alpha = 1.6732632423543772848170429916717f;
    pos = act == TMGCN_ACT_SELU ? scale : 1.f;
    neg = act == TMGCN_ACT_SELU ? scale * alpha : (act == TMGCN_ACT_LEAKY ? 0.01f : 1.f);
    use_exp = act == TMGCN_ACT_SELU;
    zero_neg = act == TMGCN_ACT_RELU;
  }
  __device__ __forceinline__ float operator()(float x) const {
    float t = x;
    if (use_exp) t = exp_nonpos(x) - 1.f;      // a uniform (scalar) branch: relu / leaky / none skip the exponential
    const float n = zero_neg ? 0.f : neg * t;
    return x > 0.f ? pos * x : n;
  }
};

// act_grad() with the activation decoded ONCE (uniform registers) instead of a switch per element: the same values,
// bit for bit — x > 0 ? pos : neg · (use_exp ? exp(x) : 1) — as selects, so a loop over elements stays branch-free
// (a branch per element also splits the loads around it: every use waits for all of them).
struct ActGrad {
  float pos, neg;
  bool use_exp;
  __device__ __forceinline__ explicit ActGrad(int act) {
    const float scale = 1.0507009873554804934193349852946f;
    const float alpha = 1.6732632423543772848170429916717f;
    pos = act == TMGCN_ACT_SELU ? scale : 1.f;
    neg = act == TMGCN_ACT_SELU ? scale * alpha : (act == TMGCN_ACT_LEAKY ? 0.01f : (act == TMGCN_ACT_RELU ? 0.f : 1.f));
    use_exp = act == TMGCN_ACT_SELU;
  }
  __device__ __forceinline__ float operator()(float x) const {
    const float e = exp_nonpos(x);
    return x > 0.f ? pos : neg * (use_exp ? e : 1.f);
  }
};

}  // namespace tmgcn

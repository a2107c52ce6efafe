// Asynchronous register staging for the stream kernels (gemm_bf16x3, mtransform_bf16x3).
//
// Problem.  A software-pipelined stream kernel wants the global loads of step n+2 in flight while it
// works on step n, across barriers and across the branches of its epilogue.  hipcc's wait-count
// pass cannot count across those branches and drains the queue (s_waitcnt vmcnt(0)) at the first
// use of a staged value, which turns a two-deep ring into a one-deep one (measured, round 1).
// Loads issued from inline asm are invisible to that pass — but if their destination is an ordinary
// asm OUTPUT the compiler believes the value is present as soon as the asm statement has "executed":
// it is free to copy it to other registers at once (a stale copy) and to reuse the original
// registers, which the memory system then overwrites when the data finally arrives.  That is not a
// theoretical concern: a first version of a staggered dW kernel (round 2, not kept) faulted exactly this way.
//
// Scheme.  The asm loads below write FIXED quads at the top of the register file (v192..v255, or
// v224..v255 for the four-quad form), declared as clobbers so that the kernel's register count
// covers them, and the kernel is written so that compiler-generated code stays below that zone:
// nothing the compiler knows about lives there.  (`amdgpu_num_vgpr` does not hard-cap the
// allocator — it was tried — so the property is CHECKED instead: tools/check_reserved_vgprs.py
// compiles the kernels to assembly and fails if any instruction outside an inline-asm block names a
// reserved register; tests/test_abi_and_host.py runs it on every CPU test run.)  After a counted
// s_waitcnt the data is moved into ordinary values with v_mov_b32 (16 moves per thread and step,
// against ~200 VALU instructions of the split it feeds).  Volatile asm statements keep their
// program order, so a TMGCN_Q_READ placed after TMGCN_WAIT_VM executes after it.
//
// Everything that uses these macros must be inlined into the kernel: the lambdas of the two kernels
// carry __attribute__((always_inline)).  Left to the inliner's size heuristics a lambda can become an
// out-of-line function (it happened when the chunked GEMM instantiations grew the kernel), in which
// the scalar base of TMGCN_Q_LOAD_S arrives in VGPRs ("invalid operand" at assembly time — loud, at
// least) and which would need its own reserved-zone check.
#pragma once

// 16-byte load into the reserved quad v[a:d]; 64-bit per-lane address
#define TMGCN_Q_LOAD(a, b, c, d, ptr) \
  asm volatile("global_load_dwordx4 v[" #a ":" #d "], %0, off" ::"v"(ptr) : "memory", "v" #a, "v" #b, "v" #c, "v" #d)
// the same with a scalar base and a 32-bit per-lane byte offset
#define TMGCN_Q_LOAD_S(a, b, c, d, voff, sbase) \
  asm volatile("global_load_dwordx4 v[" #a ":" #d "], %0, %1" ::"v"(voff), "s"(sbase) : "memory", "v" #a, "v" #b, "v" #c, "v" #d)
// move the landed quad into an ordinary 4-float value
#define TMGCN_Q_READ(a, b, c, d, dst)                                                                       \
  asm volatile("v_mov_b32 %0, v" #a "\n\tv_mov_b32 %1, v" #b "\n\tv_mov_b32 %2, v" #c "\n\tv_mov_b32 %3, v" #d \
               : "=v"((dst)[0]), "=v"((dst)[1]), "=v"((dst)[2]), "=v"((dst)[3]))
// the same, multiplied by a per-lane mask/scale on the way out (one v_mul_f32 per element instead of
// v_mov_b32 + a separate multiply: the stream kernels zero out-of-range rows / columns this way)
#define TMGCN_Q_READ_MUL(a, b, c, d, dst, z)                                                                    \
  asm volatile("v_mul_f32 %0, v" #a ", %4\n\tv_mul_f32 %1, v" #b ", %4\n\tv_mul_f32 %2, v" #c ", %4\n\tv_mul_f32 %3, v" #d ", %4" \
               : "=&v"((dst)[0]), "=&v"((dst)[1]), "=&v"((dst)[2]), "=&v"((dst)[3])                                \
               : "v"(z))
// counted wait: everything but the n youngest vector-memory operations of this wave has completed
#define TMGCN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

namespace tmgcn {

typedef float stage_f32x4 __attribute__((ext_vector_type(4)));

// Two sets (SET = 0, 1) of eight quads each in v192..v255 — for kernels whose own code stays below v192.
// I is a compile-time index; the if-constexpr ladder selects the literal register names.
template <int SET, int I>
__device__ __forceinline__ void stage8_load(const float* ptr) {
  static_assert(SET >= 0 && SET < 2 && I >= 0 && I < 8, "stage8: bad slot");
  constexpr int q = SET * 8 + I;
  if constexpr (q == 0) TMGCN_Q_LOAD(192, 193, 194, 195, ptr);
  if constexpr (q == 1) TMGCN_Q_LOAD(196, 197, 198, 199, ptr);
  if constexpr (q == 2) TMGCN_Q_LOAD(200, 201, 202, 203, ptr);
  if constexpr (q == 3) TMGCN_Q_LOAD(204, 205, 206, 207, ptr);
  if constexpr (q == 4) TMGCN_Q_LOAD(208, 209, 210, 211, ptr);
  if constexpr (q == 5) TMGCN_Q_LOAD(212, 213, 214, 215, ptr);
  if constexpr (q == 6) TMGCN_Q_LOAD(216, 217, 218, 219, ptr);
  if constexpr (q == 7) TMGCN_Q_LOAD(220, 221, 222, 223, ptr);
  if constexpr (q == 8) TMGCN_Q_LOAD(224, 225, 226, 227, ptr);
  if constexpr (q == 9) TMGCN_Q_LOAD(228, 229, 230, 231, ptr);
  if constexpr (q == 10) TMGCN_Q_LOAD(232, 233, 234, 235, ptr);
  if constexpr (q == 11) TMGCN_Q_LOAD(236, 237, 238, 239, ptr);
  if constexpr (q == 12) TMGCN_Q_LOAD(240, 241, 242, 243, ptr);
  if constexpr (q == 13) TMGCN_Q_LOAD(244, 245, 246, 247, ptr);
  if constexpr (q == 14) TMGCN_Q_LOAD(248, 249, 250, 251, ptr);
  if constexpr (q == 15) TMGCN_Q_LOAD(252, 253, 254, 255, ptr);
}

template <int SET, int I>
__device__ __forceinline__ void stage8_load_s(unsigned voff, const float* sbase) {
  static_assert(SET >= 0 && SET < 2 && I >= 0 && I < 8, "stage8: bad slot");
  constexpr int q = SET * 8 + I;
  if constexpr (q == 0) TMGCN_Q_LOAD_S(192, 193, 194, 195, voff, sbase);
  if constexpr (q == 1) TMGCN_Q_LOAD_S(196, 197, 198, 199, voff, sbase);
  if constexpr (q == 2) TMGCN_Q_LOAD_S(200, 201, 202, 203, voff, sbase);
  if constexpr (q == 3) TMGCN_Q_LOAD_S(204, 205, 206, 207, voff, sbase);
  if constexpr (q == 4) TMGCN_Q_LOAD_S(208, 209, 210, 211, voff, sbase);
  if constexpr (q == 5) TMGCN_Q_LOAD_S(212, 213, 214, 215, voff, sbase);
  if constexpr (q == 6) TMGCN_Q_LOAD_S(216, 217, 218, 219, voff, sbase);
  if constexpr (q == 7) TMGCN_Q_LOAD_S(220, 221, 222, 223, voff, sbase);
  if constexpr (q == 8) TMGCN_Q_LOAD_S(224, 225, 226, 227, voff, sbase);
  if constexpr (q == 9) TMGCN_Q_LOAD_S(228, 229, 230, 231, voff, sbase);
  if constexpr (q == 10) TMGCN_Q_LOAD_S(232, 233, 234, 235, voff, sbase);
  if constexpr (q == 11) TMGCN_Q_LOAD_S(236, 237, 238, 239, voff, sbase);
  if constexpr (q == 12) TMGCN_Q_LOAD_S(240, 241, 242, 243, voff, sbase);
  if constexpr (q == 13) TMGCN_Q_LOAD_S(244, 245, 246, 247, voff, sbase);
  if constexpr (q == 14) TMGCN_Q_LOAD_S(248, 249, 250, 251, voff, sbase);
  if constexpr (q == 15) TMGCN_Q_LOAD_S(252, 253, 254, 255, voff, sbase);
}

template <int SET, int I>
__device__ __forceinline__ stage_f32x4 stage8_read() {
  static_assert(SET >= 0 && SET < 2 && I >= 0 && I < 8, "stage8: bad slot");
  constexpr int q = SET * 8 + I;
  stage_f32x4 v;
  if constexpr (q == 0) TMGCN_Q_READ(192, 193, 194, 195, v);
  if constexpr (q == 1) TMGCN_Q_READ(196, 197, 198, 199, v);
  if constexpr (q == 2) TMGCN_Q_READ(200, 201, 202, 203, v);
  if constexpr (q == 3) TMGCN_Q_READ(204, 205, 206, 207, v);
  if constexpr (q == 4) TMGCN_Q_READ(208, 209, 210, 211, v);
  if constexpr (q == 5) TMGCN_Q_READ(212, 213, 214, 215, v);
  if constexpr (q == 6) TMGCN_Q_READ(216, 217, 218, 219, v);
  if constexpr (q == 7) TMGCN_Q_READ(220, 221, 222, 223, v);
  if constexpr (q == 8) TMGCN_Q_READ(224, 225, 226, 227, v);
  if constexpr (q == 9) TMGCN_Q_READ(228, 229, 230, 231, v);
  if constexpr (q == 10) TMGCN_Q_READ(232, 233, 234, 235, v);
  if constexpr (q == 11) TMGCN_Q_READ(236, 237, 238, 239, v);
  if constexpr (q == 12) TMGCN_Q_READ(240, 241, 242, 243, v);
  if constexpr (q == 13) TMGCN_Q_READ(244, 245, 246, 247, v);
  if constexpr (q == 14) TMGCN_Q_READ(248, 249, 250, 251, v);
  if constexpr (q == 15) TMGCN_Q_READ(252, 253, 254, 255, v);
  return v;
}

template <int SET, int I>
__device__ __forceinline__ stage_f32x4 stage8_read_mul(float z) {
  static_assert(SET >= 0 && SET < 2 && I >= 0 && I < 8, "stage8: bad slot");
  constexpr int q = SET * 8 + I;
  stage_f32x4 v;
  if constexpr (q == 0) TMGCN_Q_READ_MUL(192, 193, 194, 195, v, z);
  if constexpr (q == 1) TMGCN_Q_READ_MUL(196, 197, 198, 199, v, z);
  if constexpr (q == 2) TMGCN_Q_READ_MUL(200, 201, 202, 203, v, z);
  if constexpr (q == 3) TMGCN_Q_READ_MUL(204, 205, 206, 207, v, z);
  if constexpr (q == 4) TMGCN_Q_READ_MUL(208, 209, 210, 211, v, z);
  if constexpr (q == 5) TMGCN_Q_READ_MUL(212, 213, 214, 215, v, z);
  if constexpr (q == 6) TMGCN_Q_READ_MUL(216, 217, 218, 219, v, z);
  if constexpr (q == 7) TMGCN_Q_READ_MUL(220, 221, 222, 223, v, z);
  if constexpr (q == 8) TMGCN_Q_READ_MUL(224, 225, 226, 227, v, z);
  if constexpr (q == 9) TMGCN_Q_READ_MUL(228, 229, 230, 231, v, z);
  if constexpr (q == 10) TMGCN_Q_READ_MUL(232, 233, 234, 235, v, z);
  if constexpr (q == 11) TMGCN_Q_READ_MUL(236, 237, 238, 239, v, z);
  if constexpr (q == 12) TMGCN_Q_READ_MUL(240, 241, 242, 243, v, z);
  if constexpr (q == 13) TMGCN_Q_READ_MUL(244, 245, 246, 247, v, z);
  if constexpr (q == 14) TMGCN_Q_READ_MUL(248, 249, 250, 251, v, z);
  if constexpr (q == 15) TMGCN_Q_READ_MUL(252, 253, 254, 255, v, z);
  return v;
}

// Two sets of four quads each in v224..v255 — for kernels whose own code stays below v224 (slots 8..15 above).
template <int SET, int I>
__device__ __forceinline__ void stage4_load(const float* ptr) {
  static_assert(SET >= 0 && SET < 2 && I >= 0 && I < 4, "stage4: bad slot");
  stage8_load<1, SET * 4 + I>(ptr);
}
template <int SET, int I>
__device__ __forceinline__ stage_f32x4 stage4_read() {
  static_assert(SET >= 0 && SET < 2 && I >= 0 && I < 4, "stage4: bad slot");
  return stage8_read<1, SET * 4 + I>();
}

}  // namespace tmgcn

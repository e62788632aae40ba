// Hand-scheduled MFMA groups for the software-pipelined K loops (conv_halo.hip, wgrad_halo.hip).
//
// hipcc schedules a K-step of the halo kernels as  {ds_read x N; s_waitcnt lgkmcnt(0); 8 MFMAs; ds_read; s_waitcnt lgkmcnt(0); 8 MFMAs ..}:
// every fragment read is waited for right where it is issued, and the two waves of a SIMD run these read -> MFMA chains in lockstep
// behind the per-step barrier (SQ_VALU_MFMA_BUSY 0.53 for the 256-wide tile, profiles/r02_mfma_busy.md).  The groups below fix the
// instruction order by hand: a group is 8 (or 4) MFMAs of one weight fragment against the wave's pixel fragments with the
// ds_read_b128 of a LATER sub-step interleaved between them, and ONE counted `s_waitcnt lgkmcnt(N)` in front (LDS returns in order,
// so "at most N outstanding" names exactly which fragment has landed).  Fragments of the next sub-step are always in flight under
// the MFMAs of the current one; nothing in a group waits for a read issued inside it.
//
// Everything LDS-side of such a loop must go through these helpers: the compiler does not see the reads inside the asm statements
// and inserts no waits of its own for them (cdna_hip_programming.md 5.7); the counts are the caller's contract.
#pragma once
#include <type_traits>
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define JG_MF_(MN, i) MN " %[c" #i "], %[b], %[a" #i "], %[c" #i "]\n\t"
#define JG_RD_(k) "ds_read_b128 %[d" #k "], %[p] offset:%[o" #k "]\n\t"
#define JG_C8_(c) [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]), [c6] "+v"(c[6]), [c7] "+v"(c[7])
#define JG_A8_(a) [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7])
#define JG_C4_(c) [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3])
#define JG_A4_(a) [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3])

#define JG_G8R4_(MN)                                                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" JG_MF_(MN, 0) JG_RD_(0) JG_MF_(MN, 1) JG_MF_(MN, 2) JG_RD_(1) JG_MF_(MN, 3) JG_MF_(MN, 4)  \
               JG_RD_(2) JG_MF_(MN, 5) JG_MF_(MN, 6) JG_RD_(3) JG_MF_(MN, 7)                                                          \
               : JG_C8_(c), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)                                            \
               : JG_A8_(a), [b] "v"(b), [p] "v"(p), [w] "n"(W), [o0] "n"(O0), [o1] "n"(O1), [o2] "n"(O2), [o3] "n"(O3))
#define JG_G8R2_(MN)                                                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" JG_MF_(MN, 0) JG_MF_(MN, 1) JG_RD_(0) JG_MF_(MN, 2) JG_MF_(MN, 3) JG_MF_(MN, 4)            \
               JG_RD_(1) JG_MF_(MN, 5) JG_MF_(MN, 6) JG_MF_(MN, 7)                                                                    \
               : JG_C8_(c), [d0] "=&v"(d0), [d1] "=&v"(d1)                                                                            \
               : JG_A8_(a), [b] "v"(b), [p] "v"(p), [w] "n"(W), [o0] "n"(O0), [o1] "n"(O1))
#define JG_G8R0_(MN)                                                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" JG_MF_(MN, 0) JG_MF_(MN, 1) JG_MF_(MN, 2) JG_MF_(MN, 3) JG_MF_(MN, 4) JG_MF_(MN, 5)        \
               JG_MF_(MN, 6) JG_MF_(MN, 7)                                                                                            \
               : JG_C8_(c)                                                                                                            \
               : JG_A8_(a), [b] "v"(b), [w] "n"(W))
#define JG_G4R2_(MN)                                                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" JG_MF_(MN, 0) JG_RD_(0) JG_MF_(MN, 1) JG_MF_(MN, 2) JG_RD_(1) JG_MF_(MN, 3)               \
               : JG_C4_(c), [d0] "=&v"(d0), [d1] "=&v"(d1)                                                                            \
               : JG_A4_(a), [b] "v"(b), [p] "v"(p), [w] "n"(W), [o0] "n"(O0), [o1] "n"(O1))
#define JG_G4R0_(MN)                                                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(%[w])\n\t" JG_MF_(MN, 0) JG_MF_(MN, 1) JG_MF_(MN, 2) JG_MF_(MN, 3)                                  \
               : JG_C4_(c)                                                                                                            \
               : JG_A4_(a), [b] "v"(b), [w] "n"(W))

#define JG_BY_TYPE_(BODY)                                       \
  if constexpr (std::is_same<T, bf16_t>::value) {               \
    BODY("v_mfma_f32_16x16x32_bf16");                           \
  } else {                                                      \
    BODY("v_mfma_f32_16x16x32_f16");                            \
  }

// c[i] += b (MFMA operand A: 16 rows x 32 k) * a[i] (operand B), i = 0..7, after `lgkmcnt(W)`; reads d0..d3 <- LDS[p + O0..O3] interleaved
template <typename T, int W, int O0, int O1, int O2, int O3>
__device__ __forceinline__ void jg_g8r4(f32x4 (&c)[8], const u32x4& b, const u32x4 (&a)[8], u32x4& d0, u32x4& d1, u32x4& d2, u32x4& d3, unsigned p) {
  JG_BY_TYPE_(JG_G8R4_)
}
template <typename T, int W, int O0, int O1>
__device__ __forceinline__ void jg_g8r2(f32x4 (&c)[8], const u32x4& b, const u32x4 (&a)[8], u32x4& d0, u32x4& d1, unsigned p) {
  JG_BY_TYPE_(JG_G8R2_)
}
template <typename T, int W>
__device__ __forceinline__ void jg_g8r0(f32x4 (&c)[8], const u32x4& b, const u32x4 (&a)[8]) {
  JG_BY_TYPE_(JG_G8R0_)
}
template <typename T, int W, int O0, int O1>
__device__ __forceinline__ void jg_g4r2(f32x4 (&c)[4], const u32x4& b, const u32x4 (&a)[4], u32x4& d0, u32x4& d1, unsigned p) {
  JG_BY_TYPE_(JG_G4R2_)
}
template <typename T, int W>
__device__ __forceinline__ void jg_g4r0(f32x4 (&c)[4], const u32x4& b, const u32x4 (&a)[4]) {
  JG_BY_TYPE_(JG_G4R0_)
}

// plain fragment reads (no MFMA to hide under): 4 / 8 x ds_read_b128 from one base
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void jg_rd4(u32x4& d0, u32x4& d1, u32x4& d2, u32x4& d3, unsigned p) {
  asm volatile(JG_RD_(0) JG_RD_(1) JG_RD_(2) JG_RD_(3)
               : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
               : [p] "v"(p), [o0] "n"(O0), [o1] "n"(O1), [o2] "n"(O2), [o3] "n"(O3));
}

// ONE MFMA as an `asm volatile` statement with a memory clobber: the compiler keeps its position relative to every other volatile asm
// and to every memory access (the LDS fragment reads of a software-pipelined loop stay where the source puts them), while the operands
// remain ordinary values -- a read that feeds it is waited for by the compiler's own counted s_waitcnt.  The hazard recogniser does
// not look inside an asm statement, so the two wait states a VALU write of an operand needs in front of an MFMA (hipcc's own `s_nop 0`
// behind a v_mov of a constant fragment) are part of the statement; they issue under the previous MFMA.
template <typename T>
__device__ __forceinline__ void jg_mfma_pinned(f32x4& c, const uint4& a, const uint4& b) {
  const u32x4 av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
  if constexpr (std::is_same<T, bf16_t>::value) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv) : "memory");
  } else {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv) : "memory");
  }
}

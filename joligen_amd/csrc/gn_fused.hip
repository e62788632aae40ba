// STATUS: built, tested, measured SLOWER than the three-pass backward on every UNet shape -- NOT on the training step
// (opt-in: JG_GN_FUSED=1).  Kept as a tested kernel and as the record of the experiment; do not count it as a fusion of the step.
//
// Single-pass GroupNorm backward with the activation and its gradient RESIDENT ON CHIP between the reduction and the apply step.
//
// norm.hip runs the backward as reduce -> coef -> apply: x and dy cross HBM twice (10 B per element with the dx store).  Here one
// launch does all three: a workgroup loads its pixels of x and dy ONCE into registers (16-byte vectors, up to N per thread and tensor
// = up to 128 KB per 256-thread workgroup, three workgroups per CU: the register file is the largest on-chip store of a CU, 512 KB
// against 160 KB of LDS), accumulates the per-(image, channel) sums, adds them to `red` with agent-scope atomics, arrives on the image's
// counter, and -- once every workgroup of the IMAGE has arrived -- derives (P, Q, R) itself and writes dx straight from the registers:
// 6 B per element.  What does not fit the registers (pixels beyond N per thread) is streamed in the first phase and read again in the
// second: the kernel degrades towards the two-pass traffic instead of failing.
//
// Inter-workgroup protocol (MI355X guide, "Inter-workgroup communication"): payload = float atomics at agent scope, every wave drains
// its vmcnt before the workgroup barrier, ONE lane increments the image's arrival counter (relaxed, agent) and polls it with relaxed
// agent loads + s_sleep, ONE agent-scope acquire after the match, the sums are read back with agent-scope atomic loads.  No placement
// or dispatch-order assumption for CORRECTNESS; for PROGRESS the host keeps the workgroups of one image <= the number of CUs (every one
// of them can be resident at once whatever else shares the chip), and the spin is bounded: on expiry the kernel raises *status and goes
// on (wrong numbers, never a hang).  `red`, `cnt` must be zero at launch (rows of the executor's zero pool); *status is sticky.
#include "common.h"

namespace {

struct GnFusedArgs {
  const void* x; long ldx;
  const void* dy; long lddy; float dysc;
  const float* ab;            // [B][C][2] forward coefficients (u = a x + b)
  float* red;                 // [B][C][2] zeroed: sum du, sum du x
  unsigned* cnt;              // [B] zeroed arrival counters
  unsigned* status;           // sticky error word (spin expired)
  const float* gamma; const float* beta; const float* film; long ldfilm;
  const float* mr;            // [B][G][2] mean, rstd
  float* dgamma; float* dbeta; float* dfilm; long lddfilm;
  int G;
  void* dx; long lddx;
  const void* add1; long ldadd1; float sc1;
  const void* add2; long ldadd2; float sc2;
  int HW, C, W, K;            // K pixels per thread (K >= N: the first K - N are streamed twice)
  int dbg, sleep;             // dev ablation bits (timing only: 1 no wait, 2 no global atomics, 4 no phase-2 activation gradient), poll pause
};

struct FMap { int noct, pl, active; };
__host__ __device__ inline FMap fmap(int C) {
  FMap m;
  m.noct = C / 8;
  m.pl = 256 / m.noct;
  if (m.pl < 1) m.pl = 1;
  m.active = m.pl * m.noct;
  return m;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));       // a 16-byte vector the register allocator treats as ONE tuple
__device__ __forceinline__ uint4 as_uint4(const u32x4& v) { return make_uint4(v.x, v.y, v.z, v.w); }

constexpr unsigned kSpinLimit = 1u << 22;      // ~ seconds of s_sleep polls: only a lost workgroup gets there

// d act / d u with the hardware reciprocal (1 ulp) instead of the IEEE division of common.h's silu_grad_f: the pass is VALU-heavy
// (exp + rcp per element in BOTH phases), the division sequence alone was a third of its instructions
template <int ACT> __device__ __forceinline__ float act_grad_fast(float u) {
  if (ACT == JG_ACT_SILU) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    return s * (1.0f + u * (1.0f - s));
  }
  return act_grad_f<ACT>(u);
}

template <typename T, int ACT, bool UP, int N>
__global__ __launch_bounds__(256, (N <= 12 ? 3 : 2)) void gn_bwd_fused_kernel(const GnFusedArgs A) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];      // [2 C] channel sums | [2 G] group sums
  float* s_acc = s_mem;
  float* s_sm = s_mem + 2 * A.C;
  const int C = A.C, HW = A.HW, G = A.G, K = A.K;
  const FMap mp = fmap(C);
  const int tid = threadIdx.x, b = blockIdx.y;
  const unsigned NW = gridDim.x;
  for (int i = tid; i < 2 * C + 2 * G; i += 256) s_mem[i] = 0.f;
  const bool active = tid < mp.active;
  const int co = active ? tid % mp.noct : 0, pli = active ? tid / mp.noct : 0;
  const int pbeg = blockIdx.x * (mp.pl * K) + pli;          // this thread's first pixel of the image; its i-th is pbeg + i * pl
  // a workgroup whose K * pl pixels all lie inside the image keeps N pixels per thread resident and takes NO per-load validity test;
  // the (at most one) tail workgroup of an image streams everything twice
  const bool full = (int)(blockIdx.x + 1) * (mp.pl * K) <= HW;
  const int nstream = full ? K - N : K;
  // Addressing: a UNIFORM byte base per (tensor, pixel step) + ONE 32-bit per-thread byte offset per tensor, so that no 64-bit address
  // (nor a per-pixel validity flag) has to live in vector registers across the inter-workgroup wait.
  const char* xb = (const char*)A.x + ((long)b * HW * A.ldx) * (long)sizeof(T);
  const char* gb = (const char*)A.dy + ((long)b * (UP ? (HW >> 2) : HW) * A.lddy) * (long)sizeof(T);
  char* ob = (char*)A.dx + ((long)b * HW * A.lddx) * (long)sizeof(T);
  const unsigned vx0 = ((unsigned)pbeg * (unsigned)A.ldx + co * 8) * (unsigned)sizeof(T);
  const unsigned vg0 = ((unsigned)pbeg * (unsigned)A.lddy + co * 8) * (unsigned)sizeof(T);       // !UP only
  const unsigned vo0 = ((unsigned)pbeg * (unsigned)A.lddx + co * 8) * (unsigned)sizeof(T);
  const long sx = (long)mp.pl * A.ldx * (long)sizeof(T), sg = (long)mp.pl * A.lddy * (long)sizeof(T), so = (long)mp.pl * A.lddx * (long)sizeof(T);
  auto low_index = [&](int p) -> unsigned {        // UP: pixel of the 2x2-pooled tensor under full-resolution pixel p
    return (unsigned)((p / A.W) >> 1) * (unsigned)(A.W >> 1) + (unsigned)((p % A.W) >> 1);
  };
  // a pixel outside the image (tail workgroup) reads the thread's channel octet of pixel 0 instead -- always a valid address -- and its
  // dy is zeroed: du = 0, it adds nothing to either sum (and is never stored)
  const unsigned sx32 = (unsigned)sx, sg32 = (unsigned)sg;
  auto load_x = [&](int i, bool ok) -> u32x4 {
    const unsigned off = ok ? vx0 + (unsigned)i * sx32 : (unsigned)(co * 8 * sizeof(T));
    return *reinterpret_cast<const u32x4*>(xb + off);
  };
  auto load_g = [&](int i, int p, bool ok) -> u32x4 {
    const unsigned off = !ok ? (unsigned)(co * 8 * sizeof(T))
                             : UP ? (low_index(p) * (unsigned)A.lddy + co * 8) * (unsigned)sizeof(T) : vg0 + (unsigned)i * sg32;
    u32x4 v = *reinterpret_cast<const u32x4*>(gb + off);
    v.x = ok ? v.x : 0u;
    v.y = ok ? v.y : 0u;
    v.z = ok ? v.z : 0u;
    v.w = ok ? v.w : 0u;
    return v;
  };
  float a[8], bb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a[q] = A.ab[((long)b * C + co * 8 + q) * 2];
    bb[q] = A.ab[((long)b * C + co * 8 + q) * 2 + 1];
  }
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s1[q] = s2[q] = 0.f;
  auto accumulate = [&](const u32x4& vx, const u32x4& vg) {
    float fx[8], fg[8];
    unpack8<T>(as_uint4(vx), fx);
    unpack8<T>(as_uint4(vg), fg);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float du = UP ? fg[q] * A.dysc : fg[q];
      if (ACT != JG_ACT_NONE) du *= act_grad_fast<ACT>(a[q] * fx[q] + bb[q]);
      s1[q] += du;
      s2[q] += du * fx[q];
    }
  };
  // ---- phase 1a: pixels that do not fit the registers (streamed; read again in phase 2) ----
  for (int i = 0; i < nstream; ++i) {
    const int p = pbeg + i * mp.pl;
    const bool ok = active && p < HW;
    accumulate(load_x(i, ok), load_g(i, p, ok));           // zero-filled lanes add nothing: du = 0 * act'(b) = 0
  }
  // ---- phase 1b: the resident pixels: all loads in flight first, then the sums; the RAW vectors stay in registers ----
  u32x4 rx[N], rg[N];
  if (full && active) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      rx[j] = *reinterpret_cast<const u32x4*>(xb + (vx0 + (unsigned)(nstream + j) * sx32));
      rg[j] = UP ? *reinterpret_cast<const u32x4*>(gb + (low_index(pbeg + (nstream + j) * mp.pl) * (unsigned)A.lddy + co * 8) * (unsigned)sizeof(T))
                 : *reinterpret_cast<const u32x4*>(gb + (vg0 + (unsigned)(nstream + j) * sg32));
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      accumulate(rx[j], rg[j]);
      __builtin_amdgcn_sched_barrier(0);        // one vector pair at a time
    }
    // the raw 16-bit vectors are what stays resident: without this the compiler keeps the unpacked fp32 values (and du) of phase 1
    // alive for phase 2 (common subexpressions), i.e. 3x the registers
#pragma unroll
    for (int j = 0; j < N; ++j) {
      asm volatile("" : "+v"(rx[j]));
      asm volatile("" : "+v"(rg[j]));
    }
  }
  // ---- workgroup reduction: lanes that share a channel octet combine by xor-shuffle, then one LDS atomic per wave and channel ----
  __syncthreads();                // s_mem cleared
  if (active) {
    const bool p2 = (mp.noct & (mp.noct - 1)) == 0 && mp.noct < 64;
    if (p2) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        for (int o = mp.noct; o < 64; o <<= 1) {
          s1[q] += __shfl_xor(s1[q], o);
          s2[q] += __shfl_xor(s2[q], o);
        }
    }
    if (!p2 || (tid & 63) < mp.noct) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        atomicAdd(&s_acc[(co * 8 + q) * 2], s1[q]);
        atomicAdd(&s_acc[(co * 8 + q) * 2 + 1], s2[q]);
      }
    }
  }
  __syncthreads();
  float* redb = A.red + (long)b * 2 * C;
  if (NW > 1) {
    if (!(A.dbg & 2))
      for (int i = tid; i < 2 * C; i += 256) __hip_atomic_fetch_add(redb + i, s_acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY wave: its atomics have been performed before the arrival is published
    __syncthreads();
    if (tid == 0) {
      unsigned* cnt = A.cnt + b;
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (!(A.dbg & 1) && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NW) {
        for (int z = 0; z < A.sleep; ++z) __builtin_amdgcn_s_sleep(8);
        if (++spins > kSpinLimit) {
          __hip_atomic_store(A.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) s_acc[i] = __hip_atomic_load(redb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  // ---- coefficient step (norm.hip gn_bwd_coef_kernel) per workgroup; the first workgroup of an image emits the parameter gradients ----
  const int cpg = C / G;
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    const float gam = A.gamma ? A.gamma[c] : 1.f;
    const float f = A.film ? 1.f + A.film[(long)b * A.ldfilm + c] : 1.f;
    const float A1 = s_acc[2 * c], A2 = s_acc[2 * c + 1];
    const float mean = A.mr[((long)b * G + g) * 2], rstd = A.mr[((long)b * G + g) * 2 + 1];
    atomicAdd(&s_sm[g], gam * f * A1);
    atomicAdd(&s_sm[G + g], gam * f * rstd * (A2 - mean * A1));
    if (blockIdx.x == 0) {
      if (A.dgamma) atomicAdd(&A.dgamma[c], f * rstd * (A2 - mean * A1));
      if (A.dbeta) atomicAdd(&A.dbeta[c], f * A1);
      if (A.dfilm) {
        const float a0 = rstd * gam;
        const float b0 = (A.beta ? A.beta[c] : 0.f) - mean * a0;
        A.dfilm[(long)b * A.lddfilm + c] = a0 * A2 + b0 * A1;
        A.dfilm[(long)b * A.lddfilm + C + c] = A1;
      }
    }
  }
  __syncthreads();
  if (!active) return;
  float P[8], Q[8], R[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int c = co * 8 + q, g = c / cpg;
    const float n = (float)HW * (float)cpg;
    const float mean = A.mr[((long)b * G + g) * 2], rstd = A.mr[((long)b * G + g) * 2 + 1];
    const float M1 = s_sm[g] / n, M2 = s_sm[G + g] / n;
    const float gam = A.gamma ? A.gamma[c] : 1.f;
    const float f = A.film ? 1.f + A.film[(long)b * A.ldfilm + c] : 1.f;
    P[q] = f * gam * rstd;
    Q[q] = -rstd * rstd * M2;
    R[q] = -rstd * M1 + mean * rstd * rstd * M2;
  }
  const char* a1b = A.add1 ? (const char*)A.add1 + ((long)b * (UP ? (HW >> 2) : HW) * A.ldadd1) * (long)sizeof(T) : nullptr;
  const char* a2b = A.add2 ? (const char*)A.add2 + ((long)b * HW * A.ldadd2) * (long)sizeof(T) : nullptr;
  auto apply_store = [&](int i, int p, const u32x4& vx, const u32x4& vg) {
    float fx[8], fg[8];
    unpack8<T>(as_uint4(vx), fx);
    unpack8<T>(as_uint4(vg), fg);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float du = UP ? fg[q] * A.dysc : fg[q];
      if (ACT != JG_ACT_NONE && !(A.dbg & 4)) du *= act_grad_fast<ACT>(a[q] * fx[q] + bb[q]);
      fg[q] = du * P[q] + fx[q] * Q[q] + R[q];
    }
    if (a1b) {          // fused gradient fan-in (UP: the first addend lives at the pooled resolution like dy)
      float fa[8];
      const unsigned pa = UP ? low_index(p) : (unsigned)p;
      unpack8<T>(*reinterpret_cast<const uint4*>(a1b + ((long)pa * A.ldadd1 + co * 8) * (long)sizeof(T)), fa);
#pragma unroll
      for (int q = 0; q < 8; ++q) fg[q] += A.sc1 * fa[q];
    }
    if (a2b) {
      float fa[8];
      unpack8<T>(*reinterpret_cast<const uint4*>(a2b + ((long)p * A.ldadd2 + co * 8) * (long)sizeof(T)), fa);
#pragma unroll
      for (int q = 0; q < 8; ++q) fg[q] += A.sc2 * fa[q];
    }
    *reinterpret_cast<uint4*>(ob + (long)i * so + vo0) = pack8<T>(fg);
  };
  // ---- phase 2a: dx of the resident pixels, straight from the registers ----
  if (full) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      apply_store(nstream + j, pbeg + (nstream + j) * mp.pl, rx[j], rg[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- phase 2b: the streamed pixels again ----
  for (int i = 0; i < nstream; ++i) {
    const int p = pbeg + i * mp.pl;
    if (p < HW) apply_store(i, p, load_x(i, true), load_g(i, p, true));
  }
}

template <typename T, int ACT, bool UP>
void launch_n(int n, dim3 grid, size_t shm, hipStream_t st, const GnFusedArgs& A) {
  if (n == 8) hipLaunchKernelGGL((gn_bwd_fused_kernel<T, ACT, UP, 8>), grid, dim3(256), shm, st, A);
  else if (n == 12) hipLaunchKernelGGL((gn_bwd_fused_kernel<T, ACT, UP, 12>), grid, dim3(256), shm, st, A);
  else if (n == 20) hipLaunchKernelGGL((gn_bwd_fused_kernel<T, ACT, UP, 20>), grid, dim3(256), shm, st, A);
  else hipLaunchKernelGGL((gn_bwd_fused_kernel<T, ACT, UP, 16>), grid, dim3(256), shm, st, A);
}

}  // namespace

// GroupNorm backward in ONE launch (reduce + coef + apply of norm.hip): dx = du P + x Q + R (+ scale1 add1 + scale2 add2), dgamma / dbeta /
// dfilm as from jg_gn_bwd_coef.  `red` ([B][C][2] floats) and `counters` ([B] words) must be zero; *status (one word, sticky) is set if
// an inter-workgroup wait expired.  up != 0: dy and add1 live at the 2x2-pooled resolution (jg_gn_bwd_apply_up's addressing).
extern "C" int jg_gn_bwd_fused(int dtype, int up, const void* x, int64_t ldx, const void* dy, int64_t lddy, float dy_scale, const float* ab,
                               float* red, uint32_t* counters, uint32_t* status, const float* gamma, const float* beta, const float* film,
                               int64_t ldfilm, const float* mr, float* dgamma, float* dbeta, float* dfilm, int64_t lddfilm, int G, void* dx,
                               int64_t lddx, const void* add1, int64_t ldadd1, float scale1, const void* add2, int64_t ldadd2, float scale2,
                               int B, int H, int W, int C, int act, jg_stream_t s) {
  const int HW = H * W;
  if (!x || !dy || !ab || !red || !counters || !status || !mr || !dx) return JG_ERR_BAD_ARG;
  if (B < 1 || B > 65535 || HW < 1 || C < 8 || (C % 8) || C > 2048 || G < 1 || C % G || G > 2048) return JG_ERR_BAD_ARG;
  if (up && ((H & 1) || (W & 1))) return JG_ERR_BAD_ARG;
  if (ldx < C || lddy < C || lddx < C || (ldx % 8) || (lddy % 8) || (lddx % 8)) return JG_ERR_BAD_ARG;
  if ((add1 && (ldadd1 < C || ldadd1 % 8)) || (add2 && (ldadd2 < C || ldadd2 % 8))) return JG_ERR_BAD_ARG;
  const FMap mp = fmap(C);
  const int mode = jg_tune(JG_TUNE_GN_FUSED);            // 8 / 12 / 20: other resident depths (8, 12: three workgroups per CU; 16, 20: two); else 16
  const int n = (mode == 8 || mode == 12 || mode == 20) ? mode : 16;
  const int cap = jg_tune(JG_TUNE_GN_FUSED_CAP) > 0 ? jg_tune(JG_TUNE_GN_FUSED_CAP) : 256;   // workgroups of one image: all resident at once
  const int slots = (HW + mp.pl - 1) / mp.pl;            // pixels per thread lane over the whole image
  int K = (slots + cap - 1) / cap;
  if (K < n) K = n;
  const int nw = (slots + K - 1) / K;
  dim3 grid(nw, B);
  const size_t shm = (size_t)(2 * C + 2 * G) * sizeof(float);
  GnFusedArgs A{x, (long)ldx, dy, (long)lddy, dy_scale, ab, red, counters, status, gamma, beta, film, (long)ldfilm, mr, dgamma, dbeta, dfilm,
                (long)lddfilm, G, dx, (long)lddx, add1, (long)ldadd1, scale1, add2, (long)ldadd2, scale2, HW, C, W, K,
                jg_tune(JG_TUNE_GN_FUSED_DBG), jg_tune(JG_TUNE_GN_FUSED_SLEEP)};
  hipStream_t st = (hipStream_t)s;
  if (up) {
    JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, launch_n<T, ACT, true>(n, grid, shm, st, A);););
  } else {
    JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, launch_n<T, ACT, false>(n, grid, shm, st, A);););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}

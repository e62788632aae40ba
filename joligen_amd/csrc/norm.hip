// GroupNorm (+FiLM) (+SiLU) forward/backward for NHWC 16-bit activations, fp32 statistics.
//
// All passes are HBM-bound streaming kernels: a thread owns one 8-channel octet (one 16-byte
// vector per pixel) and a strided set of pixels, so the per-channel coefficients live in
// registers and every global access is a coalesced 16-byte load/store.
//
//   forward : stats (sum, sum^2 per (b,c))  ->  coef (a,b per (b,c))  ->  apply y = act(a x + b)
//   backward: reduce (sum du, sum du x)     ->  bwd_coef (P,Q,R)      ->  apply dx = du P + x Q + R
//
// Algorithmic bytes (T = 2 B): forward reads x twice and writes y once (6 B/elem), backward
// reads x,dy twice and writes dx once (10 B/elem).
#include "common.h"
#include <cstdlib>

// register budget of the two backward passes (experiment: -DJG_GN_BWD_WAVES=8 caps them at 64 VGPRs so that two of their waves fit a SIMD
// next to a resident weight-gradient workgroup of the side stream)
#ifdef JG_GN_BWD_WAVES
#define JG_GN_BWD_BOUNDS __launch_bounds__(256, JG_GN_BWD_WAVES)
#else
#define JG_GN_BWD_BOUNDS __launch_bounds__(256)
#endif

namespace {

struct Map {
  int noct, pl, active, chunk;
};
__host__ __device__ inline Map make_map(int C) {
  Map m;
  m.noct = C / 8;
  m.pl = 256 / m.noct;
  if (m.pl < 1) m.pl = 1;
  m.active = m.pl * m.noct;
  m.chunk = m.pl * 16;
  return m;
}

// reductions: 4x more pixels per block than the streaming passes (4x fewer global atomics per channel)
__host__ __device__ inline Map make_map_red(int C, int mult) {
  Map m = make_map(C);
  m.chunk = m.pl * mult;   // mult in {16, 32, 64}: chosen by the host so that small tensors still fill the chip
  return m;
}
inline int red_mult(int B, int HW, int C) {
  const Map m = make_map(C);
  // fewest workgroups a larger chunk must still leave (JG_GN_RED_MINBLK, A/B; see the statistics pass below: the closing global atomics of every
  // workgroup weigh more than an underfilled grid on the CUT generators' maps)
  static const int minblk = getenv("JG_GN_RED_MINBLK") ? atoi(getenv("JG_GN_RED_MINBLK")) : 2048;
  for (int mult = 64; mult > 16; mult >>= 1)
    if ((long)B * ((HW + m.pl * mult - 1) / (m.pl * mult)) >= minblk) return mult;
  return 16;
}

// Deterministic forms (JG_DETERMINISTIC 1, `det`): ONE workgroup per image walks all its pixels (no cross-workgroup atomics), and the
// threads that share a channel octet are combined by an ORDERED sum over their LDS slots instead of ds_add_f32 chains: the result does not
// depend on the order in which waves and workgroups happen to run.  s_part: 256 threads x 16 partial sums behind the [C][2] row.
__device__ __forceinline__ void ordered_octet_sum(float* s_acc, const float (&s1)[8], const float (&s2)[8], const Map& mp, int C, bool active) {
  float* s_part = s_acc + 2 * C;
  const int tid = threadIdx.x;
  if (active) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      s_part[tid * 16 + q * 2] = s1[q];
      s_part[tid * 16 + q * 2 + 1] = s2[q];
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) {
    const int c = i >> 1, k = i & 1, co = c >> 3, q = c & 7;
    float s = 0.f;
    for (int pl = 0; pl < mp.pl; ++pl) s += s_part[(pl * mp.noct + co) * 16 + q * 2 + k];
    s_acc[i] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, long ldx, float* __restrict__ sums, long ldsums,
                                                       int HW, int C, int det, int chunk) {
  extern __shared__ float s_acc[];  // [C][2] (+ [256][16] in the deterministic form)
  Map mp = make_map(C);
  mp.chunk = chunk;
  const int tid = threadIdx.x, b = blockIdx.y;
  for (int i = tid; i < 2 * C; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s1[q] = s2[q] = 0.f;
  if (tid < mp.active) {
    const int co = tid % mp.noct, pl = tid / mp.noct;
    const int pbeg = det ? 0 : blockIdx.x * mp.chunk;
    const int pend = det ? HW : min(HW, pbeg + mp.chunk);
    const T* xb = x + ((long)b * HW) * ldx + co * 8;
    // four pixels per trip, their loads issued together (round 6; as in gn_bwd_reduce_kernel: one 16-byte load in flight per wave left the
    // InstanceNorm statistics passes of the CUT generators at 40 us for 8 us of bytes)
    int p = pbeg + pl;
    for (; p + 3 * mp.pl < pend; p += 4 * mp.pl) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(xb + (long)(p + u * mp.pl) * ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8<T>(v[u], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          s1[q] += f[q];
          s2[q] += f[q] * f[q];
        }
      }
    }
    for (; p < pend; p += mp.pl) {
      const uint4 v = *reinterpret_cast<const uint4*>(xb + (long)p * ldx);
      float f[8];
      unpack8<T>(v, f);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        s1[q] += f[q];
        s2[q] += f[q] * f[q];
      }
    }
    if (!det) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        atomicAdd(&s_acc[(co * 8 + q) * 2], s1[q]);
        atomicAdd(&s_acc[(co * 8 + q) * 2 + 1], s2[q]);
      }
    }
  }
  if (det) ordered_octet_sum(s_acc, s1, s2, mp, C, tid < mp.active);      // every thread of the workgroup: it holds a barrier
  __syncthreads();
  // (deterministic form: one workgroup per image, so ONE add per address onto a value that is itself reproducible)
  for (int i = tid; i < 2 * C; i += 256) atomicAdd(&sums[(long)b * 2 * ldsums + i], s_acc[i]);
}

// One 16-lane group per (b, c): the lanes split the (slot, channel-of-group) statistics between them and
// combine with xor-shuffles (the slot loop is nslots * cpg long: serial it took 17 us per call).
__global__ __launch_bounds__(256) void gn_coef_kernel(const float* __restrict__ sums, long ldsums, int nslots,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ film, long ldfilm, float* __restrict__ ab,
                                                      float* __restrict__ mr, int B, int HW, int C, int G, float eps) {
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  if (idx >= B * C) return;   // whole 16-lane groups leave together
  const int b = idx / C, c = idx % C;
  const int cpg = C / G, g = c / cpg;
  float S1 = 0.f, S2 = 0.f;
  const int total = nslots * cpg;
  for (int i = sub; i < total; i += 16) {
    const int sl = i / cpg, k = i - sl * cpg;
    const long o = (((long)b * nslots + sl) * ldsums + g * cpg + k) * 2;
    S1 += sums[o];
    S2 += sums[o + 1];
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    S1 += __shfl_xor(S1, o);
    S2 += __shfl_xor(S2, o);
  }
  if (sub != 0) return;
  const float n = (float)HW * (float)cpg;
  const float mean = S1 / n;
  const float var = fmaxf(S2 / n - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  const float a0 = rstd * (gamma ? gamma[c] : 1.f);
  const float b0 = (beta ? beta[c] : 0.f) - mean * a0;
  float a = a0, bb = b0;
  if (film) {
    const float sc = film[(long)b * ldfilm + c], sh = film[(long)b * ldfilm + C + c];
    a = a0 * (1.f + sc);
    bb = b0 * (1.f + sc) + sh;
  }
  ab[(long)idx * 2] = a;
  ab[(long)idx * 2 + 1] = bb;
  if (c == g * cpg) {
    mr[((long)b * G + g) * 2] = mean;
    mr[((long)b * G + g) * 2 + 1] = rstd;
  }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ ab,
                                                       T* __restrict__ y, long ldy, int HW, int C, const T* __restrict__ add = nullptr, long ldadd = 0,
                                                       int rev = 0) {
  // add (round 6, jg_gn_apply_add): y = act(a x + b) + add -- the residual sum `x + conv_block(x)` of a ResnetBlock, whose branch ends in an
  // InstanceNorm (resnet_generator.py:11-95), formed in the norm's apply pass instead of by a sum kernel behind it
  const Map mp = make_map(C);
  // rev (JG_GN_REVERSE bit 2, A/B): from the END -- the producing convolution's last tiles may still be in the Infinity Cache, and the
  // consumer behind (a convolution that starts at image 0) finds what this pass wrote last
  const int tid = threadIdx.x, b = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
  if (tid >= mp.active) return;
  const int co = tid % mp.noct, pl = tid / mp.noct;
  float a[8], bb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a[q] = ab[((long)b * C + co * 8 + q) * 2];
    bb[q] = ab[((long)b * C + co * 8 + q) * 2 + 1];
  }
  const int pbeg = (rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * mp.chunk;
  const int pend = min(HW, pbeg + mp.chunk);
  const long row0 = (long)b * HW;
  auto one = [&](const uint4& v, int p) {
    float f[8];
    unpack8<T>(v, f);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float u = a[q] * f[q] + bb[q];
      f[q] = act_f<ACT>(u);
    }
    if (add) {          // the ROUNDED norm output plus the addend, as the separate sum kernel computed it
      float r8[8], g8[8];
      unpack8<T>(pack8<T>(f), g8);
      unpack8<T>(*reinterpret_cast<const uint4*>(add + (row0 + p) * ldadd + co * 8), r8);
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] = g8[q] + r8[q];
    }
    *reinterpret_cast<uint4*>(y + (row0 + p) * ldy + co * 8) = pack8<T>(f);
  };
  // four pixels per trip, their loads issued together (one load in flight per wave otherwise: see gn_bwd_reduce_kernel)
  int p = pbeg + pl;
  for (; p + 3 * mp.pl < pend; p += 4 * mp.pl) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(x + (row0 + p + u * mp.pl) * ldx + co * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(v[u], p + u * mp.pl);
  }
  for (; p < pend; p += mp.pl) one(*reinterpret_cast<const uint4*>(x + (row0 + p) * ldx + co * 8), p);
}

// y[B, H/2, W/2, C] = scale * sum over the 2x2 window of act(a * x + b): the ResBlock-down path's pool(act(norm(x))) without the
// full-resolution intermediate.  HWo = (H/2)*(W/2), Wo = W/2
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_apply_pool_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ ab,
                                                            T* __restrict__ y, long ldy, int HWo, int Wo, int C, float scale) {
  const Map mp = make_map(C);
  const int tid = threadIdx.x, b = blockIdx.y;
  if (tid >= mp.active) return;
  const int co = tid % mp.noct, pl = tid / mp.noct;
  float a[8], bb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    a[q] = ab[((long)b * C + co * 8 + q) * 2];
    bb[q] = ab[((long)b * C + co * 8 + q) * 2 + 1];
  }
  const int pbeg = blockIdx.x * mp.chunk;
  const int pend = min(HWo, pbeg + mp.chunk);
  const long W = 2L * Wo;
  for (int p = pbeg + pl; p < pend; p += mp.pl) {
    const int ho = p / Wo, wo = p - ho * Wo;
    const T* src = x + ((long)b * HWo * 4 + 2L * ho * W + 2 * wo) * ldx + co * 8;
    uint4 v[4];
    v[0] = *reinterpret_cast<const uint4*>(src);
    v[1] = *reinterpret_cast<const uint4*>(src + ldx);
    v[2] = *reinterpret_cast<const uint4*>(src + W * ldx);
    v[3] = *reinterpret_cast<const uint4*>(src + (W + 1) * ldx);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float f[8];
      unpack8<T>(v[i], f);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += act_f<ACT>(a[q] * f[q] + bb[q]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] *= scale;
    *reinterpret_cast<uint4*>(y + ((long)b * HWo + p) * ldy + co * 8) = pack8<T>(acc);
  }
}

// UP: dy is the gradient of a 2x2 average pool's OUTPUT ([B, H/2, W/2, C], W = full-resolution width): the pool's adjoint
// (nearest upsample * dysc) is applied while reading, the full-resolution gradient is never materialised
template <typename T, int ACT, bool UP = false>
__global__ JG_GN_BWD_BOUNDS void gn_bwd_reduce_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy,
                                                            long lddy, const float* __restrict__ ab,
                                                            float* __restrict__ red, int HW, int C, int mult, int W = 0,
                                                            float dysc = 1.f, int det = 0, int rev = 0) {
  extern __shared__ float s_acc[];  // [C][2] (+ [256][16] in the deterministic form, see gn_stats_kernel)
  const Map mp = make_map_red(C, mult);
  // rev (JG_GN_REVERSE bit 1, A/B): walk from the END -- the tail of dy that the producing convolution wrote last may still sit in the
  // 256 MB Infinity Cache; the apply pass behind then walks forward (bit 0 clear)
  const int tid = threadIdx.x, b = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
  for (int i = tid; i < 2 * C; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  float s1[8], s2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s1[q] = s2[q] = 0.f;
  if (tid < mp.active) {
    const int co = tid % mp.noct, pl = tid / mp.noct;
    float a[8], bb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a[q] = ab[((long)b * C + co * 8 + q) * 2];
      bb[q] = ab[((long)b * C + co * 8 + q) * 2 + 1];
    }
    const int pbeg = det ? 0 : (rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * mp.chunk;
    const int pend = det ? HW : min(HW, pbeg + mp.chunk);
    const long row0 = (long)b * HW;
    auto dy_row = [&](int p) -> long {
      return UP ? ((long)b * (HW >> 2) + (long)((p / W) >> 1) * (W >> 1) + ((p % W) >> 1)) : row0 + p;
    };
    auto accumulate = [&](const uint4& vx, const uint4& vg) {
      float fx[8], fg[8];
      unpack8<T>(vx, fx);
      unpack8<T>(vg, fg);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float du = UP ? fg[q] * dysc : fg[q];
        if (ACT != JG_ACT_NONE) du *= act_grad_f<ACT>(a[q] * fx[q] + bb[q]);
        s1[q] += du;
        s2[q] += du * fx[q];
      }
    };
    // four pixels per trip with all eight 16-byte loads issued before the first use.  (`#pragma unroll 4` alone left every pair of
    // loads next to its arithmetic: two loads in flight per wave, 32 KB per CU, which at ~2 us of latency is the 4.1 TB/s this
    // read-only pass measured against 5.3 TB/s of the apply pass -- profiles/r04_pmc_hbm_traffic.md.)
    int p = pbeg + pl;
    for (; p + 3 * mp.pl < pend; p += 4 * mp.pl) {
      uint4 vx[4], vg[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        vx[u] = *reinterpret_cast<const uint4*>(x + (row0 + p + u * mp.pl) * ldx + co * 8);
        vg[u] = *reinterpret_cast<const uint4*>(dy + dy_row(p + u * mp.pl) * lddy + co * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accumulate(vx[u], vg[u]);
    }
    for (; p < pend; p += mp.pl) {
      const uint4 vx = *reinterpret_cast<const uint4*>(x + (row0 + p) * ldx + co * 8);
      const uint4 vg = *reinterpret_cast<const uint4*>(dy + dy_row(p) * lddy + co * 8);
      accumulate(vx, vg);
    }
    // lanes of a wave that share the channel octet (tid % noct, noct a power of two < 64) combine by xor-shuffle
    // first: one LDS atomic per wave and channel instead of up to 32 colliding ones
    const bool p2 = (mp.noct & (mp.noct - 1)) == 0 && mp.noct < 64;
    if (!det) {
      if (p2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          for (int o = mp.noct; o < 64; o <<= 1) {
            s1[q] += __shfl_xor(s1[q], o);
            s2[q] += __shfl_xor(s2[q], o);
          }
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
  }
  if (det) ordered_octet_sum(s_acc, s1, s2, mp, C, tid < mp.active);
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) atomicAdd(&red[(long)b * 2 * C + i], s_acc[i]);
}

__device__ __forceinline__ float red_sum(const float* __restrict__ red, int nslots, int b, int C, int c, int j) {
  float s = 0.f;
  for (int sl = 0; sl < nslots; ++sl) s += red[(((long)b * nslots + sl) * C + c) * 2 + j];
  return s;
}

__global__ void gn_bwd_coef_kernel(const float* __restrict__ red, int nslots, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const float* __restrict__ film, long ldfilm,
                                   const float* __restrict__ mr, float* __restrict__ pqr, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta, float* __restrict__ dfilm, long lddfilm, int B, int HW,
                                   int C, int G) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * C) return;
  const int b = idx / C, c = idx % C;
  const int cpg = C / G, g = c / cpg;
  const float mean = mr[((long)b * G + g) * 2], rstd = mr[((long)b * G + g) * 2 + 1];
  float SM1 = 0.f, SM2 = 0.f;
  for (int k = 0; k < cpg; ++k) {
    const int cc = g * cpg + k;
    const float gam = gamma ? gamma[cc] : 1.f;
    const float f = film ? 1.f + film[(long)b * ldfilm + cc] : 1.f;
    const float A1 = red_sum(red, nslots, b, C, cc, 0), A2 = red_sum(red, nslots, b, C, cc, 1);
    SM1 += gam * f * A1;
    SM2 += gam * f * rstd * (A2 - mean * A1);
  }
  const float n = (float)HW * (float)cpg;
  const float M1 = SM1 / n, M2 = SM2 / n;
  const float gam = gamma ? gamma[c] : 1.f;
  const float f = film ? 1.f + film[(long)b * ldfilm + c] : 1.f;
  const float A1 = red_sum(red, nslots, b, C, c, 0), A2 = red_sum(red, nslots, b, C, c, 1);
  pqr[(long)idx * 3] = f * gam * rstd;
  pqr[(long)idx * 3 + 1] = -rstd * rstd * M2;
  pqr[(long)idx * 3 + 2] = -rstd * M1 + mean * rstd * rstd * M2;
  if (dgamma) atomicAdd(&dgamma[c], f * rstd * (A2 - mean * A1));
  if (dbeta) atomicAdd(&dbeta[c], f * A1);
  if (dfilm) {
    const float a0 = rstd * gam;
    const float b0 = (beta ? beta[c] : 0.f) - mean * a0;
    dfilm[(long)b * lddfilm + c] = a0 * A2 + b0 * A1;
    dfilm[(long)b * lddfilm + C + c] = A1;
  }
}

// Deterministic form of the parameter gradients (JG_DETERMINISTIC 1): one thread per channel adds the images' terms IN ORDER and writes its
// sum once (gn_bwd_coef_kernel is then launched without dgamma / dbeta: its per-(image, channel) atomics come in any order).
__global__ void gn_bwd_param_det_kernel(const float* __restrict__ red, int nslots, const float* __restrict__ film, long ldfilm,
                                        const float* __restrict__ mr, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, int G) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int cpg = C / G, g = c / cpg;
  float dg = 0.f, db = 0.f;
  for (int b = 0; b < B; ++b) {
    const float mean = mr[((long)b * G + g) * 2], rstd = mr[((long)b * G + g) * 2 + 1];
    const float f = film ? 1.f + film[(long)b * ldfilm + c] : 1.f;
    const float A1 = red_sum(red, nslots, b, C, c, 0), A2 = red_sum(red, nslots, b, C, c, 1);
    dg += f * rstd * (A2 - mean * A1);
    db += f * A1;
  }
  if (dgamma) dgamma[c] += dg;
  if (dbeta) dbeta[c] += db;
}

// FC ("fused coefficients"): the pass derives its own (P, Q, R) from the reductions instead of reading them from a coefficient kernel's
// output -- what gn_bwd_coef_kernel computes per (image, channel), done once per workgroup for the image it works on (2 C floats of `red`
// through LDS); the workgroup of pixel chunk 0 also emits the parameter / FiLM gradients.  One launch fewer per GroupNorm backward: next to
// the weight-gradient stream a 6 us coefficient kernel waited ~50 us for a free slot (3 ms of every step).
struct GnFc {
  const float* red;      // [B][C][2]: sum du, sum du x
  const float* gamma;
  const float* beta;
  const float* film;     // [B][ldfilm]: scale | shift, or null
  const float* mr;       // [B][G][2]: mean, rstd
  float* dgamma;
  float* dbeta;
  float* dfilm;
  long ldfilm, lddfilm;
  int G;
};

// UP: dy AND add1 are low-resolution ([B, H/2, W/2, C]) and read through the nearest-upsample index map (see gn_bwd_reduce_kernel)
template <typename T, int ACT, bool UP = false, bool FC = false>
__global__ JG_GN_BWD_BOUNDS void gn_bwd_apply_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy,
                                                           long lddy, const float* __restrict__ ab,
                                                           const float* __restrict__ pqr, T* __restrict__ dx, long lddx,
                                                           const T* __restrict__ add1, long ldadd1, float sc1,
                                                           const T* __restrict__ add2, long ldadd2, float sc2, int HW,
                                                           int C, int rev, int W = 0, float dysc = 1.f, GnFc fc = GnFc()) {
  __shared__ float s_sm[FC ? 2 * 256 : 2];      // group sums SM1 | SM2 (G <= 256)
  const Map mp = make_map(C);
  // reversed traversal: the pass runs right behind gn_bwd_reduce over the same (x, dy); walking from the END
  // re-reads what the reduction touched last and is still resident in the 256 MB Infinity Cache
  const int tid = threadIdx.x, b = rev ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
  if (FC) {
    const int G = fc.G, cpg = C / G;
    for (int i = tid; i < 2 * G; i += 256) s_sm[i] = 0.f;
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      const int g = c / cpg;
      const float gam = fc.gamma ? fc.gamma[c] : 1.f;
      const float f = fc.film ? 1.f + fc.film[(long)b * fc.ldfilm + c] : 1.f;
      const float A1 = fc.red[((long)b * C + c) * 2], A2 = fc.red[((long)b * C + c) * 2 + 1];
      const float mean = fc.mr[((long)b * G + g) * 2], rstd = fc.mr[((long)b * G + g) * 2 + 1];
      atomicAdd(&s_sm[g], gam * f * A1);
      atomicAdd(&s_sm[G + g], gam * f * rstd * (A2 - mean * A1));
      if (blockIdx.x == 0) {       // once per image: parameter and FiLM gradients (gn_bwd_coef_kernel's side outputs)
        if (fc.dgamma) atomicAdd(&fc.dgamma[c], f * rstd * (A2 - mean * A1));
        if (fc.dbeta) atomicAdd(&fc.dbeta[c], f * A1);
        if (fc.dfilm) {
          const float a0 = rstd * gam;
          const float b0 = (fc.beta ? fc.beta[c] : 0.f) - mean * a0;
          fc.dfilm[(long)b * fc.lddfilm + c] = a0 * A2 + b0 * A1;
          fc.dfilm[(long)b * fc.lddfilm + C + c] = A1;
        }
      }
    }
    __syncthreads();
  }
  if (tid >= mp.active) return;
  const int co = tid % mp.noct, pl = tid / mp.noct;
  float a[8], bb[8], P[8], Q[8], R[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const long o = (long)b * C + co * 8 + q;
    a[q] = ab[o * 2];
    bb[q] = ab[o * 2 + 1];
    if (FC) {
      const int c = co * 8 + q, G = fc.G, cpg = C / G, g = c / cpg;
      const float n = (float)HW * (float)cpg;
      const float mean = fc.mr[((long)b * G + g) * 2], rstd = fc.mr[((long)b * G + g) * 2 + 1];
      const float M1 = s_sm[g] / n, M2 = s_sm[G + g] / n;
      const float gam = fc.gamma ? fc.gamma[c] : 1.f;
      const float f = fc.film ? 1.f + fc.film[(long)b * fc.ldfilm + c] : 1.f;
      P[q] = f * gam * rstd;
      Q[q] = -rstd * rstd * M2;
      R[q] = -rstd * M1 + mean * rstd * rstd * M2;
    } else {
      P[q] = pqr[o * 3];
      Q[q] = pqr[o * 3 + 1];
      R[q] = pqr[o * 3 + 2];
    }
  }
  const int pbeg = (rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * mp.chunk;
  const int pend = min(HW, pbeg + mp.chunk);
  const long row0 = (long)b * HW;
  auto dy_row = [&](int p) -> long {
    return UP ? ((long)b * (HW >> 2) + (long)((p / W) >> 1) * (W >> 1) + ((p % W) >> 1)) : row0 + p;
  };
  auto one = [&](const uint4& vx, const uint4& vg, const uint4& va1, const uint4& va2, int p) {
    float fx[8], fg[8];
    unpack8<T>(vx, fx);
    unpack8<T>(vg, fg);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float du = UP ? fg[q] * dysc : fg[q];
      if (ACT != JG_ACT_NONE) du *= act_grad_f<ACT>(a[q] * fx[q] + bb[q]);
      fg[q] = du * P[q] + fx[q] * Q[q] + R[q];
    }
    if (add1) {   // fused gradient accumulation of the other consumers of x (residual / skip / concat paths)
      float fa[8];
      unpack8<T>(va1, fa);
#pragma unroll
      for (int q = 0; q < 8; ++q) fg[q] += sc1 * fa[q];
    }
    if (add2) {
      float fa[8];
      unpack8<T>(va2, fa);
#pragma unroll
      for (int q = 0; q < 8; ++q) fg[q] += sc2 * fa[q];
    }
    *reinterpret_cast<uint4*>(dx + (row0 + p) * lddx + co * 8) = pack8<T>(fg);
  };
  // NB pixels per trip: their 2 NB to 4 NB loads are issued together (see gn_bwd_reduce_kernel).  NB = 4 (137 VGPRs, an occupancy step
  // down) measured 0.5 ms/step SLOWER than 2 on the palette step (profiles/r04_gn_load_batching_ab.log)
#ifdef JG_GN_APPLY_NB
  constexpr int NB = JG_GN_APPLY_NB;
#else
  constexpr int NB = 2;
#endif
  int p = pbeg + pl;
  for (; p + (NB - 1) * mp.pl < pend; p += NB * mp.pl) {
    uint4 vx[NB], vg[NB], va1[NB], va2[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int pp = p + u * mp.pl;
      const long pg = dy_row(pp);
      vx[u] = *reinterpret_cast<const uint4*>(x + (row0 + pp) * ldx + co * 8);
      vg[u] = *reinterpret_cast<const uint4*>(dy + pg * lddy + co * 8);
      va1[u] = add1 ? *reinterpret_cast<const uint4*>(add1 + pg * ldadd1 + co * 8) : make_uint4(0, 0, 0, 0);
      va2[u] = add2 ? *reinterpret_cast<const uint4*>(add2 + (row0 + pp) * ldadd2 + co * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) one(vx[u], vg[u], va1[u], va2[u], p + u * mp.pl);
  }
  for (; p < pend; p += mp.pl) {
    const long pg = dy_row(p);
    one(*reinterpret_cast<const uint4*>(x + (row0 + p) * ldx + co * 8), *reinterpret_cast<const uint4*>(dy + pg * lddy + co * 8),
        add1 ? *reinterpret_cast<const uint4*>(add1 + pg * ldadd1 + co * 8) : make_uint4(0, 0, 0, 0),
        add2 ? *reinterpret_cast<const uint4*>(add2 + (row0 + p) * ldadd2 + co * 8) : make_uint4(0, 0, 0, 0), p);
  }
}

inline bool bad_shape(int B, int HW, int C) { return B < 1 || HW < 1 || C < 8 || (C % 8) || C > 2048 || B > 65535; }

}  // namespace

extern "C" int jg_gn_stats_ld(int dtype, const void* x, int64_t ldx, float* sums, int64_t ldsums, int B, int HW, int C,
                              jg_stream_t s) {
  if (!x || !sums || bad_shape(B, HW, C) || ldx < C || (ldx % 8) || ldsums < C) return JG_ERR_BAD_ARG;
  Map mp = make_map(C);
  {      // pixels per thread of a statistics workgroup.  Round 6: 64 instead of 16 wherever that still leaves >= 256 workgroups -- every workgroup ends
         // with 2 C global atomics, and on the CUT generators' maps (64 x 64 x 256 channels x 32 images) those, not the bytes, set the time:
         // mobile_resnet_attn + [projected_d, basic] 370 -> 380 images/s, resnet + basic 476 -> 486 (8: 355, 32: 377, 128: 376, 256: 365); the
         // UNet step does not change.  JG_GN_STATS_MULT overrides (A/B).
    static const int forced = getenv("JG_GN_STATS_MULT") ? atoi(getenv("JG_GN_STATS_MULT")) : 0;
    int mult = 16;
    for (int m = 64; m > 16; m >>= 1)
      if ((long)B * ((HW + mp.pl * m - 1) / (mp.pl * m)) >= 256) { mult = m; break; }
    if (forced >= 4 && forced <= 256) mult = forced;
    mp.chunk = mp.pl * mult;
  }
  const int det = jg_tune(JG_TUNE_DETERMINISTIC) != 0;
  dim3 grid(det ? 1 : (HW + mp.chunk - 1) / mp.chunk, B);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gn_stats_kernel<T>), grid, dim3(256), (2 * C + (det ? 4096 : 0)) * sizeof(float), (hipStream_t)s,
                                              (const T*)x, (long)ldx, sums, (long)ldsums, HW, C, det, mp.chunk););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_stats(int dtype, const void* x, float* sums, int B, int HW, int C, jg_stream_t s) {
  if (!x || !sums || bad_shape(B, HW, C)) return JG_ERR_BAD_ARG;
  if (hipMemsetAsync(sums, 0, sizeof(float) * 2 * B * C, (hipStream_t)s) != hipSuccess) return JG_ERR_LAUNCH;
  return jg_gn_stats_ld(dtype, x, C, sums, C, B, HW, C, s);
}

extern "C" int jg_gn_coef_ld(const float* sums, int64_t ldsums, int nslots, const float* gamma, const float* beta,
                             const float* film, int64_t ldfilm, float* ab, float* mr, int B, int HW, int C, int G, float eps,
                             jg_stream_t s) {
  if (!sums || !ab || !mr || G < 1 || C % G || ldsums < C || nslots < 1) return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(gn_coef_kernel, dim3((B * C + 15) / 16), dim3(256), 0, (hipStream_t)s, sums, (long)ldsums, nslots, gamma, beta,
                     film, (long)ldfilm, ab, mr, B, HW, C, G, eps);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_coef(const float* sums, const float* gamma, const float* beta, const float* film, int64_t ldfilm,
                          float* ab, float* mr, int B, int HW, int C, int G, float eps, jg_stream_t s) {
  return jg_gn_coef_ld(sums, C, 1, gamma, beta, film, ldfilm, ab, mr, B, HW, C, G, eps, s);
}

extern "C" int jg_gn_apply_ld(int dtype, const void* x, int64_t ldx, const float* ab, void* y, int64_t ldy, int B, int HW,
                              int C, int act, jg_stream_t s) {
  if (!x || !ab || !y || bad_shape(B, HW, C) || ldx < C || ldy < C || (ldx % 8) || (ldy % 8)) return JG_ERR_BAD_ARG;
  const Map mp = make_map(C);
  dim3 grid((HW + mp.chunk - 1) / mp.chunk, B);
  hipStream_t st = (hipStream_t)s;
  JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_apply_kernel<T, ACT>), grid, dim3(256), 0, st, (const T*)x,
                                                            (long)ldx, ab, (T*)y, (long)ldy, HW, C, (const T*)nullptr, 0L,
                                                            (jg_tune(JG_TUNE_GN_REVERSE) >> 2) & 1);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_apply_pool(int dtype, const void* x, int64_t ldx, const float* ab, void* y, int64_t ldy, int B, int H, int W, int C,
                                int act, float scale, jg_stream_t s) {
  if (!x || !ab || !y || bad_shape(B, H * W, C) || (H & 1) || (W & 1) || ldx < C || ldy < C || (ldx % 8) || (ldy % 8)) return JG_ERR_BAD_ARG;
  const int HWo = (H / 2) * (W / 2);
  const Map mp = make_map(C);
  dim3 grid((HWo + mp.chunk - 1) / mp.chunk, B);
  JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_apply_pool_kernel<T, ACT>), grid, dim3(256), 0, (hipStream_t)s,
                                                            (const T*)x, (long)ldx, ab, (T*)y, (long)ldy, HWo, W / 2, C, scale);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_apply_add(int dtype, const void* x, int64_t ldx, const float* ab, const void* add, int64_t ldadd, void* y, int64_t ldy, int B,
                               int HW, int C, int act, jg_stream_t s) {
  if (!x || !ab || !y || !add || bad_shape(B, HW, C) || ldx < C || ldy < C || ldadd < C || (ldx % 8) || (ldy % 8) || (ldadd % 8)) return JG_ERR_BAD_ARG;
  const Map mp = make_map(C);
  dim3 grid((HW + mp.chunk - 1) / mp.chunk, B);
  hipStream_t st = (hipStream_t)s;
  JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_apply_kernel<T, ACT>), grid, dim3(256), 0, st, (const T*)x,
                                                            (long)ldx, ab, (T*)y, (long)ldy, HW, C, (const T*)add, (long)ldadd);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_apply(int dtype, const void* x, const float* ab, void* y, int B, int HW, int C, int act,
                           jg_stream_t s) {
  return jg_gn_apply_ld(dtype, x, C, ab, y, C, B, HW, C, act, s);
}

static int gn_bwd_reduce_ld_impl(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* ab,
                                 float* red, int B, int HW, int C, int act, jg_stream_t s, bool zero) {
  if (!x || !dy || !ab || !red || bad_shape(B, HW, C) || ldx < C || lddy < C || (ldx % 8) || (lddy % 8)) return JG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  if (zero && hipMemsetAsync(red, 0, sizeof(float) * 2 * B * C, st) != hipSuccess) return JG_ERR_LAUNCH;
  const int mult = red_mult(B, HW, C);
  const Map mp = make_map_red(C, mult);
  const int det = jg_tune(JG_TUNE_DETERMINISTIC) != 0;
  dim3 grid(det ? 1 : (HW + mp.chunk - 1) / mp.chunk, B);
  const size_t shm = (2 * C + (det ? 4096 : 0)) * sizeof(float);
  JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, ACT>), grid, dim3(256), shm, st, (const T*)x,
                                                            (long)ldx, (const T*)dy, (long)lddy, ab, red, HW, C, mult, 0, 1.f, det, (jg_tune(JG_TUNE_GN_REVERSE) >> 1) & 1);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_bwd_reduce_ld(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* ab,
                                   float* red, int B, int HW, int C, int act, jg_stream_t s) {
  return gn_bwd_reduce_ld_impl(dtype, x, ldx, dy, lddy, ab, red, B, HW, C, act, s, true);
}
// the same, ACCUMULATING into `red` (the caller hands over zeroed rows of a pool it clears once per backward pass)
extern "C" int jg_gn_bwd_reduce_ld_acc(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* ab,
                                       float* red, int B, int HW, int C, int act, jg_stream_t s) {
  return gn_bwd_reduce_ld_impl(dtype, x, ldx, dy, lddy, ab, red, B, HW, C, act, s, false);
}

static int gn_bwd_reduce_up_impl(int dtype, const void* x, int64_t ldx, const void* dy_low, int64_t lddy, float dy_scale, const float* ab,
                                 float* red, int B, int H, int W, int C, int act, jg_stream_t s, bool zero) {
  const int HW = H * W;
  if (!x || !dy_low || !ab || !red || bad_shape(B, HW, C) || (H & 1) || (W & 1) || ldx < C || lddy < C || (ldx % 8) || (lddy % 8)) return JG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  if (zero && hipMemsetAsync(red, 0, sizeof(float) * 2 * B * C, st) != hipSuccess) return JG_ERR_LAUNCH;
  const int mult = red_mult(B, HW, C);
  const Map mp = make_map_red(C, mult);
  const int det = jg_tune(JG_TUNE_DETERMINISTIC) != 0;
  dim3 grid(det ? 1 : (HW + mp.chunk - 1) / mp.chunk, B);
  const size_t shm = (2 * C + (det ? 4096 : 0)) * sizeof(float);
  JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, ACT, true>), grid, dim3(256), shm, st, (const T*)x,
                                                            (long)ldx, (const T*)dy_low, (long)lddy, ab, red, HW, C, mult, W, dy_scale, det, (jg_tune(JG_TUNE_GN_REVERSE) >> 1) & 1);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_bwd_reduce_up(int dtype, const void* x, int64_t ldx, const void* dy_low, int64_t lddy, float dy_scale, const float* ab,
                                   float* red, int B, int H, int W, int C, int act, jg_stream_t s) {
  return gn_bwd_reduce_up_impl(dtype, x, ldx, dy_low, lddy, dy_scale, ab, red, B, H, W, C, act, s, true);
}
extern "C" int jg_gn_bwd_reduce_up_acc(int dtype, const void* x, int64_t ldx, const void* dy_low, int64_t lddy, float dy_scale,
                                       const float* ab, float* red, int B, int H, int W, int C, int act, jg_stream_t s) {
  return gn_bwd_reduce_up_impl(dtype, x, ldx, dy_low, lddy, dy_scale, ab, red, B, H, W, C, act, s, false);
}

extern "C" int jg_gn_bwd_apply_up(int dtype, const void* x, int64_t ldx, const void* dy_low, int64_t lddy, float dy_scale, const float* ab,
                                  const float* pqr, void* dx, int64_t lddx, const void* add1_low, int64_t ldadd1, float scale1,
                                  const void* add2, int64_t ldadd2, float scale2, int B, int H, int W, int C, int act, jg_stream_t s) {
  const int HW = H * W;
  if (!x || !dy_low || !ab || !pqr || !dx || bad_shape(B, HW, C) || (H & 1) || (W & 1)) return JG_ERR_BAD_ARG;
  if (ldx < C || lddy < C || lddx < C || (ldx % 8) || (lddy % 8) || (lddx % 8)) return JG_ERR_BAD_ARG;
  if ((add1_low && (ldadd1 < C || ldadd1 % 8)) || (add2 && (ldadd2 < C || ldadd2 % 8))) return JG_ERR_BAD_ARG;
  const Map mp = make_map(C);
  dim3 grid((HW + mp.chunk - 1) / mp.chunk, B);
  const int rev = jg_tune(JG_TUNE_GN_REVERSE) & 1;
  JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_bwd_apply_kernel<T, ACT, true>), grid, dim3(256), 0, (hipStream_t)s,
                                                            (const T*)x, (long)ldx, (const T*)dy_low, (long)lddy, ab, pqr, (T*)dx, (long)lddx,
                                                            (const T*)add1_low, (long)ldadd1, scale1, (const T*)add2, (long)ldadd2, scale2,
                                                            HW, C, rev, W, dy_scale);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

// GroupNorm backward, second pass WITH its coefficient step (no jg_gn_bwd_coef launch): dx = du P + x Q + R (+ addends) with (P, Q, R) derived
// in the kernel from `red` (one slot), gamma / beta / FiLM and (mean, rstd); dgamma / dbeta / dfilm are produced as by jg_gn_bwd_coef.
// up != 0: dy (and add1) live at the 2x2-pooled resolution (jg_gn_bwd_apply_up's addressing, dy_scale applied on read).
extern "C" int jg_gn_bwd_apply_fc(int dtype, int up, const void* x, int64_t ldx, const void* dy, int64_t lddy, float dy_scale, const float* ab,
                                  const float* red, const float* gamma, const float* beta, const float* film, int64_t ldfilm, const float* mr,
                                  float* dgamma, float* dbeta, float* dfilm, int64_t lddfilm, int G, void* dx, int64_t lddx, const void* add1,
                                  int64_t ldadd1, float scale1, const void* add2, int64_t ldadd2, float scale2, int B, int H, int W, int C,
                                  int act, jg_stream_t s) {
  const int HW = H * W;
  if (!x || !dy || !ab || !red || !mr || !dx || bad_shape(B, HW, C) || G < 1 || G > 256 || C % G) return JG_ERR_BAD_ARG;
  if (up && ((H & 1) || (W & 1))) return JG_ERR_BAD_ARG;
  if (ldx < C || lddy < C || lddx < C || (ldx % 8) || (lddy % 8) || (lddx % 8)) return JG_ERR_BAD_ARG;
  if ((add1 && (ldadd1 < C || ldadd1 % 8)) || (add2 && (ldadd2 < C || ldadd2 % 8))) return JG_ERR_BAD_ARG;
  const Map mp = make_map(C);
  dim3 grid((HW + mp.chunk - 1) / mp.chunk, B);
  const int rev = jg_tune(JG_TUNE_GN_REVERSE) & 1;
  GnFc fc{red, gamma, beta, film, mr, dgamma, dbeta, dfilm, (long)ldfilm, (long)lddfilm, G};
  if (up) {
    JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_bwd_apply_kernel<T, ACT, true, true>), grid, dim3(256), 0, (hipStream_t)s,
                                                              (const T*)x, (long)ldx, (const T*)dy, (long)lddy, ab, (const float*)nullptr, (T*)dx,
                                                              (long)lddx, (const T*)add1, (long)ldadd1, scale1, (const T*)add2, (long)ldadd2,
                                                              scale2, HW, C, rev, W, dy_scale, fc);););
  } else {
    JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_bwd_apply_kernel<T, ACT, false, true>), grid, dim3(256), 0, (hipStream_t)s,
                                                              (const T*)x, (long)ldx, (const T*)dy, (long)lddy, ab, (const float*)nullptr, (T*)dx,
                                                              (long)lddx, (const T*)add1, (long)ldadd1, scale1, (const T*)add2, (long)ldadd2,
                                                              scale2, HW, C, rev, 0, 1.f, fc);););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_bwd_reduce(int dtype, const void* x, const void* dy, const float* ab, float* red, int B, int HW,
                                int C, int act, jg_stream_t s) {
  return jg_gn_bwd_reduce_ld(dtype, x, C, dy, C, ab, red, B, HW, C, act, s);
}

extern "C" int jg_gn_bwd_coef_slots(const float* red, int nslots, const float* gamma, const float* beta, const float* film,
                                    int64_t ldfilm, const float* mr, float* pqr, float* dgamma, float* dbeta, float* dfilm,
                                    int64_t lddfilm, int B, int HW, int C, int G, jg_stream_t s) {
  if (!red || !mr || !pqr || G < 1 || C % G || nslots < 1) return JG_ERR_BAD_ARG;
  const bool det = jg_tune(JG_TUNE_DETERMINISTIC) != 0 && (dgamma || dbeta);
  hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)s, red, nslots, gamma, beta, film,
                     (long)ldfilm, mr, pqr, det ? nullptr : dgamma, det ? nullptr : dbeta, dfilm, (long)lddfilm, B, HW, C, G);
  if (det)
    hipLaunchKernelGGL(gn_bwd_param_det_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)s, red, nslots, film, (long)ldfilm, mr, dgamma, dbeta,
                       B, C, G);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_bwd_coef(const float* red, const float* gamma, const float* beta, const float* film,
                              int64_t ldfilm, const float* mr, float* pqr, float* dgamma, float* dbeta, float* dfilm,
                              int64_t lddfilm, int B, int HW, int C, int G, jg_stream_t s) {
  return jg_gn_bwd_coef_slots(red, 1, gamma, beta, film, ldfilm, mr, pqr, dgamma, dbeta, dfilm, lddfilm, B, HW, C, G, s);
}

extern "C" int jg_gn_bwd_apply_ld(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* ab,
                                  const float* pqr, void* dx, int64_t lddx, const void* add1, int64_t ldadd1, float scale1,
                                  const void* add2, int64_t ldadd2, float scale2, int B, int HW, int C, int act,
                                  jg_stream_t s) {
  if (!x || !dy || !ab || !pqr || !dx || bad_shape(B, HW, C)) return JG_ERR_BAD_ARG;
  if (ldx < C || lddy < C || lddx < C || (ldx % 8) || (lddy % 8) || (lddx % 8)) return JG_ERR_BAD_ARG;
  if ((add1 && (ldadd1 < C || ldadd1 % 8)) || (add2 && (ldadd2 < C || ldadd2 % 8))) return JG_ERR_BAD_ARG;
  const Map mp = make_map(C);
  dim3 grid((HW + mp.chunk - 1) / mp.chunk, B);
  hipStream_t st = (hipStream_t)s;
  const int rev = jg_tune(JG_TUNE_GN_REVERSE) & 1;
  JG_DISPATCH_DTYPE(dtype, JG_DISPATCH_ACT(act, hipLaunchKernelGGL((gn_bwd_apply_kernel<T, ACT>), grid, dim3(256), 0, st, (const T*)x,
                                                            (long)ldx, (const T*)dy, (long)lddy, ab, pqr, (T*)dx, (long)lddx,
                                                            (const T*)add1, (long)ldadd1, scale1, (const T*)add2, (long)ldadd2,
                                                            scale2, HW, C, rev);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gn_bwd_apply(int dtype, const void* x, const void* dy, const float* ab, const float* pqr, void* dx,
                               int B, int HW, int C, int act, jg_stream_t s) {
  return jg_gn_bwd_apply_ld(dtype, x, C, dy, C, ab, pqr, dx, C, nullptr, 0, 0.f, nullptr, 0, 0.f, B, HW, C, act, s);
}

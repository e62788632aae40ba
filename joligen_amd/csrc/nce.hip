// Patch-contrastive losses of the CUT path, fp32:
//   * jg_sgemm            strided/batched fp32 GEMM (PatchSampleF MLP, the q k^T similarity bmm, their adjoints, and
//                          the small nn.Linear layers of the embedding path) -- LDS-tiled 64x64x32 on v_mfma_f32_32x32x2_f32
//   * jg_nce_sinkhorn_fwd  MoNCE optimal-transport weights: K = exp(S), 50 Sinkhorn iterations, one workgroup per image
//   * jg_nce_ce            cross-entropy over [l_pos | l_neg] / T per patch, loss + dS (+ dW for MoNCE) in one pass
//   * jg_nce_sinkhorn_bwd  reverse sweep through the Sinkhorn iterations (the reference differentiates through them w.r.t. q)
//   * jg_nce_sinkhorn_gk   dS += K .* (rank-2T update assembled from the forward/backward histories)
// reference: models/modules/NCE/base_NCE.py:17-77, monce.py:16-33, sinkhorn.py:6-58
#include <stdlib.h>

#include "common.h"

namespace {

struct SgemmP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* E;
  int M, N, K;
  long sam, sak, sbn, sbk, scm, scn;
  long ba, bb, bc;
  float alpha, beta;
  int act_a, act_b, act_e;
  int ksplit;   // > 1: blockIdx.z is a K slice (single batch), the epilogue is atomicAdd(alpha * acc) onto C
};

__device__ __forceinline__ float act_rt(float v, int act) {
  return act == JG_ACT_SILU ? v / (1.0f + expf(-v)) : act == JG_ACT_RELU ? fmaxf(v, 0.f) : v;
}
__device__ __forceinline__ float act_grad_rt2(float v, int act) {
  if (act == JG_ACT_RELU) return v > 0.f ? 1.0f : 0.f;
  if (act != JG_ACT_SILU) return 1.0f;
  const float s = 1.0f / (1.0f + expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

// C[z][m][n] = alpha * sum_k actA(A[z][m][k]) actB(B[z][n][k]) + bias[n], then *= act'(E[z][m][n]), then += beta * C.
// AKC / BKC: the operand is contiguous along k (else along m / n); picks the coalesced load pattern.
//
// 64 x 64 output tile per workgroup, K in steps of 32 through LDS (k-major: As[kk][m], Bs[kk][n]); each of the 4 waves owns one
// 32 x 32 quadrant and accumulates it with v_mfma_f32_32x32x2_f32 (true fp32 multiply-add, 16 accumulator registers per lane):
// operand A of the instruction = lane (row l % 32, k l / 32), operand B = lane (k l / 32, column l % 32) -- one ds_read_b32 each, bank
// conflict free in the k-major layout.  Against the 4 x 4 register-tile FMA loop this replaced: 1/8 of the LDS operand traffic, no
// VALU issue in the inner loop, 16-byte global loads where the operand is contiguous and aligned (K steps of 32 instead of 16).
typedef float sg_f32x16 __attribute__((ext_vector_type(16)));
constexpr int SG_BK = 32;

// 4 consecutive elements along the contiguous axis (stride 1) starting at `p`, or guarded scalars
__device__ __forceinline__ void sg_load4(const float* __restrict__ p, bool vec, const bool (&ok)[4], long stride, float (&v)[4]) {
  if (vec) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ok[i] ? p[i * stride] : 0.f;
  }
}

// a 64 (rows) x 32 (k) operand tile: global -> 8 registers per thread (sg_fetch), registers -> S[kk][row] (sg_put); KC: contiguous
// along k.  Split so that the loads of tile k + 1 are in flight while the MFMAs of tile k run.
template <bool KC>
__device__ __forceinline__ void sg_fetch(const float* __restrict__ X, long srow, long sk, int nrows, int row0, int k0, int kend, int t,
                                         float (&v)[8]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // KC: row t >> 2, k = (t & 3) * 8 + h * 4 .. + 3;   else: rows (t & 15) * 4 .. + 3, k = (t >> 4) + h * 16
    const int r = KC ? (t >> 2) : (t & 15) * 4;
    const int kk = KC ? (t & 3) * 8 + h * 4 : (t >> 4) + h * 16;
    const float* src = X + (long)(row0 + r) * srow + (long)(k0 + kk) * sk;
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ok[i] = KC ? (row0 + r < nrows && k0 + kk + i < kend) : (k0 + kk < kend && row0 + r + i < nrows);
    const long st = KC ? sk : srow;
    const bool vec = ok[3] && st == 1 && ((reinterpret_cast<size_t>(src) & 15) == 0);
    float q[4];
    sg_load4(src, vec, ok, st, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[h * 4 + i] = q[i];
  }
}
template <bool KC>
__device__ __forceinline__ void sg_put(const float (&v)[8], int act, float (*S)[68], int t) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (KC) {
      const int r = t >> 2, kk = (t & 3) * 8 + h * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) S[kk + i][r] = act_rt(v[h * 4 + i], act);
    } else {
      const int r = (t & 15) * 4, kk = (t >> 4) + h * 16;
      *reinterpret_cast<float4*>(&S[kk][r]) =
          make_float4(act_rt(v[h * 4], act), act_rt(v[h * 4 + 1], act), act_rt(v[h * 4 + 2], act), act_rt(v[h * 4 + 3], act));
    }
  }
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void sgemm_kernel(SgemmP p) {
  __shared__ float As[SG_BK][68];
  __shared__ float Bs[SG_BK][68];
  const int z = p.ksplit > 1 ? 0 : blockIdx.z;
  const int kchunk = p.ksplit > 1 ? ((p.K + p.ksplit - 1) / p.ksplit + SG_BK - 1) / SG_BK * SG_BK : p.K;
  const int kbeg = p.ksplit > 1 ? blockIdx.z * kchunk : 0;
  const int kend = min(p.K, kbeg + kchunk);
  const float* A = p.A + z * p.ba;
  const float* B = p.B + z * p.bb;
  float* C = p.C + z * p.bc;
  const float* E = p.E ? p.E + z * p.bc : nullptr;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int qm = (wave >> 1) * 32, qn = (wave & 1) * 32;      // this wave's quadrant of the tile
  const int l32 = lane & 31, lk = lane >> 5;
  sg_f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  float ra[8], rb[8];
  sg_fetch<AKC>(A, p.sam, p.sak, p.M, m0, kbeg, kend, t, ra);
  sg_fetch<BKC>(B, p.sbn, p.sbk, p.N, n0, kbeg, kend, t, rb);
  for (int k0 = kbeg; k0 < kend; k0 += SG_BK) {
    sg_put<AKC>(ra, p.act_a, As, t);
    sg_put<BKC>(rb, p.act_b, Bs, t);
    __syncthreads();
    if (k0 + SG_BK < kend) {      // next tile's global loads fly over this tile's MFMAs
      sg_fetch<AKC>(A, p.sam, p.sak, p.M, m0, k0 + SG_BK, kend, t, ra);
      sg_fetch<BKC>(B, p.sbn, p.sbk, p.N, n0, k0 + SG_BK, kend, t, rb);
    }
#pragma unroll
    for (int kk = 0; kk < SG_BK; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + lk][qm + l32], Bs[kk + lk][qn + l32], acc, 0, 0, 0);
    __syncthreads();
  }
  // accumulator element i of lane l: row 8 * (i / 4) + 4 * (l / 32) + i % 4, column l % 32 of the quadrant
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + qm + 8 * (i >> 2) + 4 * lk + (i & 3);
    const int n = n0 + qn + l32;
    if (m >= p.M || n >= p.N) continue;
    const long o = (long)m * p.scm + (long)n * p.scn;
    float c = p.alpha * acc[i];
    if (p.ksplit > 1) {
      atomicAdd(&C[o], c);
      continue;
    }
    if (p.bias) c += p.bias[n];
    if (E) c *= act_grad_rt2(E[o], p.act_e);
    if (p.beta != 0.f) c += p.beta * C[o];
    C[o] = c;
  }
}

// ---- the same GEMM on the 16-bit matrix cores with SPLIT fp32 operands (round 6) ---------------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits per operand, and a . b ~ hi_a hi_b + hi_a lo_b + lo_a hi_b (the lo lo
// term is 2^-16 of the product) in fp32 accumulators: relative error ~ 2^-16 per product against 2^-24 of v_mfma_f32_32x32x2_f32 -- and three
// v_mfma_f32_16x16x32_bf16 per 32-deep step instead of sixteen fp32 MFMAs at 1/16 of the rate: 5 x the MFMA throughput.  The PatchSampleF
// GEMMs and the q k^T products of the contrastive losses (16 K rows x 256 x 256) took 26 - 35 us each at 125 TFLOP/s, the fp32 roofline; their
// inputs are features of a 16-bit encoder.  MEASURED: 35.7 -> 24.3 us alone (16384 x 256 x 256), the CUT steps unchanged (20.90 vs 20.86 ms,
// 34.36 vs 34.26 ms: profiles/r06_sgemm_split_ab.txt) -- these launches are not what the main queue waits for; opt-in (JG_SGEMM_SPLIT=1), since it
// moves results by 1e-5 for nothing.  Same tile (64 x 64, K steps of 32), same operand loaders and epilogue as sgemm_kernel; LDS holds
// the two halves row-major [row][k] (80-byte rows: the 16-byte fragment reads of 16 rows fall into distinct bank groups).
constexpr int SG3_LD = 40;      // bf16 elements per LDS row (32 + 8 of padding)
__device__ __forceinline__ void sg_split(float x, uint16_t& hi, uint16_t& lo) {
  const bf16_t h = from_f32<bf16_t>(x);
  hi = to_bits<bf16_t>(h);
  lo = to_bits<bf16_t>(from_f32<bf16_t>(x - to_f32(h)));
}
template <bool KC>
__device__ __forceinline__ void sg3_put(const float (&v)[8], int act, uint16_t (*Sh)[SG3_LD], uint16_t (*Sl)[SG3_LD], int t) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (KC) {
      const int r = t >> 2, kk = (t & 3) * 8 + h * 4;
      uint16_t hi[4], lo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sg_split(act_rt(v[h * 4 + i], act), hi[i], lo[i]);
      *reinterpret_cast<uint2*>(&Sh[r][kk]) = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
      *reinterpret_cast<uint2*>(&Sl[r][kk]) = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
    } else {
      const int r = (t & 15) * 4, kk = (t >> 4) + h * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint16_t hi, lo;
        sg_split(act_rt(v[h * 4 + i], act), hi, lo);
        Sh[r + i][kk] = hi;
        Sl[r + i][kk] = lo;
      }
    }
  }
}
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256) void sgemm3_kernel(SgemmP p) {
  __shared__ __attribute__((aligned(16))) uint16_t Ah[64][SG3_LD];
  __shared__ __attribute__((aligned(16))) uint16_t Al[64][SG3_LD];
  __shared__ __attribute__((aligned(16))) uint16_t Bh[64][SG3_LD];
  __shared__ __attribute__((aligned(16))) uint16_t Bl[64][SG3_LD];
  const int z = p.ksplit > 1 ? 0 : blockIdx.z;
  const int kchunk = p.ksplit > 1 ? ((p.K + p.ksplit - 1) / p.ksplit + SG_BK - 1) / SG_BK * SG_BK : p.K;
  const int kbeg = p.ksplit > 1 ? blockIdx.z * kchunk : 0;
  const int kend = min(p.K, kbeg + kchunk);
  const float* A = p.A + z * p.ba;
  const float* B = p.B + z * p.bb;
  float* C = p.C + z * p.bc;
  const float* E = p.E ? p.E + z * p.bc : nullptr;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int qm = (wave >> 1) * 32, qn = (wave & 1) * 32;
  const int l16 = lane & 15, kg = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float ra[8], rb[8];
  sg_fetch<AKC>(A, p.sam, p.sak, p.M, m0, kbeg, kend, t, ra);
  sg_fetch<BKC>(B, p.sbn, p.sbk, p.N, n0, kbeg, kend, t, rb);
  for (int k0 = kbeg; k0 < kend; k0 += SG_BK) {
    sg3_put<AKC>(ra, p.act_a, Ah, Al, t);
    sg3_put<BKC>(rb, p.act_b, Bh, Bl, t);
    __syncthreads();
    if (k0 + SG_BK < kend) {
      sg_fetch<AKC>(A, p.sam, p.sak, p.M, m0, k0 + SG_BK, kend, t, ra);
      sg_fetch<BKC>(B, p.sbn, p.sbk, p.N, n0, k0 + SG_BK, kend, t, rb);
    }
    uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = *reinterpret_cast<const uint4*>(&Ah[qm + i * 16 + l16][kg * 8]);
      al[i] = *reinterpret_cast<const uint4*>(&Al[qm + i * 16 + l16][kg * 8]);
      bh[i] = *reinterpret_cast<const uint4*>(&Bh[qn + i * 16 + l16][kg * 8]);
      bl[i] = *reinterpret_cast<const uint4*>(&Bl[qn + i * 16 + l16][kg * 8]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = Mfma<bf16_t>::run(al[i], bh[j], acc[i][j]);      // the small terms first
        acc[i][j] = Mfma<bf16_t>::run(ah[i], bl[j], acc[i][j]);
        acc[i][j] = Mfma<bf16_t>::run(ah[i], bh[j], acc[i][j]);
      }
    __syncthreads();
  }
  // D of a 16 x 16 tile: row = 4 * (lane / 16) + q, column = lane % 16
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + qm + i * 16 + kg * 4 + q;
        const int n = n0 + qn + j * 16 + l16;
        if (m >= p.M || n >= p.N) continue;
        const long o = (long)m * p.scm + (long)n * p.scn;
        float c = p.alpha * acc[i][j][q];
        if (p.ksplit > 1) {
          atomicAdd(&C[o], c);
          continue;
        }
        if (p.bias) c += p.bias[n];
        if (E) c *= act_grad_rt2(E[o], p.act_e);
        if (p.beta != 0.f) c += p.beta * C[o];
        C[o] = c;
      }
}

int sgemm_launch(const SgemmP& p, int nbatch, hipStream_t st) {
  const dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, p.ksplit > 1 ? p.ksplit : nbatch);
  const bool akc = p.sak == 1, bkc = p.sbk == 1;
  if (jg_tune(JG_TUNE_SGEMM_SPLIT) != 0) {        // opt-in: split-bf16 operands on the 16-bit matrix cores (above); default: v_mfma_f32_32x32x2_f32
    if (akc && bkc) hipLaunchKernelGGL((sgemm3_kernel<true, true>), grid, dim3(256), 0, st, p);
    else if (akc) hipLaunchKernelGGL((sgemm3_kernel<true, false>), grid, dim3(256), 0, st, p);
    else if (bkc) hipLaunchKernelGGL((sgemm3_kernel<false, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((sgemm3_kernel<false, false>), grid, dim3(256), 0, st, p);
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  if (akc && bkc) hipLaunchKernelGGL((sgemm_kernel<true, true>), grid, dim3(256), 0, st, p);
  else if (akc) hipLaunchKernelGGL((sgemm_kernel<true, false>), grid, dim3(256), 0, st, p);
  else if (bkc) hipLaunchKernelGGL((sgemm_kernel<false, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((sgemm_kernel<false, false>), grid, dim3(256), 0, st, p);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

// ---- Sinkhorn (sinkhorn.py:6-24) -------------------------------------------------------------------------------------
// one 1024-thread workgroup per image; K [P,P] lives in global memory (L2-resident: 256 KB at P = 256), u / v in LDS.
constexpr int SK_THREADS = 1024;
constexpr int SK_MAXP = 1024;

// out[i] = sum_j K[i][j] vec[j]  (wave per row, lanes along j)
__device__ __forceinline__ void row_matvec(const float* __restrict__ K, const float* vec, float* out, int P, int wave, int lane,
                                           int nwaves) {
  for (int i = wave; i < P; i += nwaves) {
    float acc = 0.f;
    for (int j = lane; j < P; j += 64) acc += K[(long)i * P + j] * vec[j];
    acc = wave_sum(acc);
    if (lane == 0) out[i] = acc;
  }
}
// out[j] = sum_i vec[i] K[i][j]  (threads along j, row groups reduced through LDS); contains its own barriers
__device__ __forceinline__ void col_matvec(const float* __restrict__ K, const float* vec, float* out, float* part, int P, int t) {
  const int ppad = (P + 63) & ~63;
  const int ngroups = SK_THREADS / ppad < 1 ? 1 : SK_THREADS / ppad;
  const int j = t % ppad, grp = t / ppad;
  if (grp < ngroups && j < P) {
    float acc = 0.f;
    for (int i = grp; i < P; i += ngroups) acc += vec[i] * K[(long)i * P + j];
    part[grp * ppad + j] = acc;
  }
  __syncthreads();
  if (t < P) {
    float acc = 0.f;
    for (int g = 0; g < ngroups; ++g) acc += part[g * ppad + t];
    out[t] = acc;
  }
  __syncthreads();
}

__global__ __launch_bounds__(SK_THREADS) void sinkhorn_fwd_kernel(const float* __restrict__ S, float* __restrict__ Kout,
                                                                   float* __restrict__ u_hist, float* __restrict__ v_hist, int P,
                                                                   int niter, float eps) {
  __shared__ float s_u[SK_MAXP], s_v[SK_MAXP], s_tmp[SK_MAXP], s_part[SK_THREADS];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* Sb = S + (long)b * P * P;
  float* K = Kout + (long)b * P * P;
  float* uh = u_hist + (long)b * niter * P;
  float* vh = v_hist + (long)b * (niter + 1) * P;
  for (long i = t; i < (long)P * P; i += SK_THREADS) {
    const int r = i / P, c = i % P;
    K[i] = expf((r == c ? -10.0f : Sb[i]) / eps);
  }
  for (int i = t; i < P; i += SK_THREADS) {
    s_v[i] = 1.0f;
    vh[i] = 1.0f;
  }
  __threadfence_block();
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    row_matvec(K, s_v, s_tmp, P, wave, lane, SK_THREADS / 64);
    __syncthreads();
    for (int i = t; i < P; i += SK_THREADS) {
      const float u = 1.0f / s_tmp[i];  // a = out_size / in_size = 1
      s_u[i] = u;
      uh[(long)it * P + i] = u;
    }
    __syncthreads();
    col_matvec(K, s_u, s_tmp, s_part, P, t);
    for (int i = t; i < P; i += SK_THREADS) {
      const float v = 1.0f / s_tmp[i];
      s_v[i] = v;
      vh[(long)(it + 1) * P + i] = v;
    }
    __syncthreads();
  }
}

// reverse sweep: inputs gW = dL/d(u_i K_ij v_j); outputs the per-iteration adjoints ds_t, dr_t
__global__ __launch_bounds__(SK_THREADS) void sinkhorn_bwd_kernel(const float* __restrict__ Kin, const float* __restrict__ u_hist,
                                                                   const float* __restrict__ v_hist, float* __restrict__ gW,
                                                                   float* __restrict__ ds_hist, float* __restrict__ dr_hist, int P,
                                                                   int niter) {
  __shared__ float s_gu[SK_MAXP], s_gv[SK_MAXP], s_x[SK_MAXP], s_tmp[SK_MAXP], s_part[SK_THREADS];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* K = Kin + (long)b * P * P;
  float* G = gW + (long)b * P * P;
  const float* uh = u_hist + (long)b * niter * P;
  const float* vh = v_hist + (long)b * (niter + 1) * P;
  float* dsh = ds_hist + (long)b * niter * P;
  float* drh = dr_hist + (long)b * niter * P;
  // G <- gW .* K in place (both seeds contract against it)
  for (long i = t; i < (long)P * P; i += SK_THREADS) G[i] *= K[i];
  for (int i = t; i < P; i += SK_THREADS) {
    s_x[i] = vh[(long)niter * P + i];
    s_tmp[i] = uh[(long)(niter - 1) * P + i];
  }
  __threadfence_block();
  __syncthreads();
  row_matvec(G, s_x, s_gu, P, wave, lane, SK_THREADS / 64);   // gu_i = sum_j gW_ij K_ij v_j
  __syncthreads();
  col_matvec(G, s_tmp, s_gv, s_part, P, t);                   // gv_j = sum_i gW_ij u_i K_ij
  for (int it = niter - 1; it >= 0; --it) {
    // v_{it+1} = 1 / (K^T u_it):  ds = -gv v^2 ; gu += K ds
    for (int i = t; i < P; i += SK_THREADS) {
      const float v = vh[(long)(it + 1) * P + i];
      const float ds = -s_gv[i] * v * v;
      s_x[i] = ds;
      dsh[(long)it * P + i] = ds;
    }
    __syncthreads();
    row_matvec(K, s_x, s_tmp, P, wave, lane, SK_THREADS / 64);
    __syncthreads();
    // u_it = 1 / (K v_it):  dr = -gu u^2 ; gv <- K^T dr ; gu <- 0
    for (int i = t; i < P; i += SK_THREADS) {
      const float u = uh[(long)it * P + i];
      const float dr = -(s_gu[i] + s_tmp[i]) * u * u;
      s_x[i] = dr;
      drh[(long)it * P + i] = dr;
      s_gu[i] = 0.f;
    }
    __syncthreads();
    col_matvec(K, s_x, s_gv, s_part, P, t);
  }
}

// ---- register-resident Sinkhorn for P <= 256 -------------------------------------------------------------------------------
// 1024 threads as a 32 x 32 grid of 8 x 8 tiles of K (64 VGPRs per lane): both matvecs of an iteration run out of registers.
// Lane (ti, tj) = (t >> 5, t & 31); a wave holds 2 tile rows x 32 tile columns.
//   row matvec  y_i = sum_j K_ij x_j : 8 partial sums per lane, reduced over the 32 tj lanes by a halving exchange (9 shuffles)
//   col matvec  z_j = sum_i x_i K_ij : 8 partial sums per lane, halved over the 2 ti of the wave, then 16 waves through LDS
constexpr int SKR = 256;

struct SkTile {
  float k[8][8];
};

// after the call, lanes with (lane & 3) == 0 hold the row sum of tile row  a = ((lane >> 2) & 7)  (bits 4,3,2 of the lane)
__device__ __forceinline__ float sk_row_reduce(const float* p, int lane) {
  float k4[4], k2[2];
  const bool h16 = lane & 16, h8 = lane & 8, h4 = lane & 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float recv = __shfl_xor(h16 ? p[k] : p[k + 4], 16);
    k4[k] = (h16 ? p[k + 4] : p[k]) + recv;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float recv = __shfl_xor(h8 ? k4[k] : k4[k + 2], 8);
    k2[k] = (h8 ? k4[k + 2] : k4[k]) + recv;
  }
  float r = (h4 ? k2[1] : k2[0]) + __shfl_xor(h4 ? k2[0] : k2[1], 4);
  r += __shfl_xor(r, 2);
  r += __shfl_xor(r, 1);
  return r;
}
__device__ __forceinline__ int sk_row_of(int t, int lane) { return 8 * (t >> 5) + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1); }

__device__ __forceinline__ void sk_row_partials(const SkTile& K, const float* __restrict__ vec, int tj, float* p) {
  const float4 a = *reinterpret_cast<const float4*>(vec + 8 * tj);
  const float4 b = *reinterpret_cast<const float4*>(vec + 8 * tj + 4);
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += K.k[r][c] * x[c];
    p[r] = acc;
  }
}
// halve the 8 column partials over the wave's two tile rows and park them in part[wave][256]
__device__ __forceinline__ void sk_col_store(const float* q, int tj, int lane, float* part_w) {
  const bool h32 = lane & 32;
  float4 o;
  float k4[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float recv = __shfl_xor(h32 ? q[k] : q[k + 4], 32);
    k4[k] = (h32 ? q[k + 4] : q[k]) + recv;
  }
  o.x = k4[0]; o.y = k4[1]; o.z = k4[2]; o.w = k4[3];
  *reinterpret_cast<float4*>(part_w + 8 * tj + (h32 ? 4 : 0)) = o;
}
// partial column sums of this wave -> part[wave][256]
__device__ __forceinline__ void sk_col_partials(const SkTile& K, const float* __restrict__ vec, int ti, int tj, int lane, float* part_w) {
  const float4 a = *reinterpret_cast<const float4*>(vec + 8 * ti);
  const float4 b = *reinterpret_cast<const float4*>(vec + 8 * ti + 4);
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  float q[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) acc += x[r] * K.k[r][c];
    q[c] = acc;
  }
  sk_col_store(q, tj, lane, part_w);
}

__global__ __launch_bounds__(1024) void sinkhorn_fwd_reg_kernel(const float* __restrict__ S, float* __restrict__ Kout,
                                                                 float* __restrict__ u_hist, float* __restrict__ v_hist, int P, int niter,
                                                                 float eps) {
  __shared__ __attribute__((aligned(16))) float s_u[SKR], s_v[SKR], s_part[16][SKR];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, ti = t >> 5, tj = t & 31;
  const float* Sb = S + (long)b * P * P;
  float* Kb = Kout + (long)b * P * P;
  float* uh = u_hist + (long)b * niter * P;
  float* vh = v_hist + (long)b * (niter + 1) * P;
  SkTile K;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = 8 * ti + r;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int col = 8 * tj + c;
      const bool ok = row < P && col < P;
      const float v = ok ? expf((row == col ? -10.0f : Sb[(long)row * P + col]) / eps) : 0.f;
      K.k[r][c] = v;
      if (ok) Kb[(long)row * P + col] = v;
    }
  }
  if (t < SKR) {
    s_v[t] = t < P ? 1.0f : 0.f;
    if (t < P) vh[t] = 1.0f;
  }
  __syncthreads();
  const int myrow = sk_row_of(t, lane);
  for (int it = 0; it < niter; ++it) {
    float p[8];
    sk_row_partials(K, s_v, tj, p);
    const float y = sk_row_reduce(p, lane);
    if ((lane & 3) == 0) {
      const float u = myrow < P ? 1.0f / y : 0.f;
      s_u[myrow] = u;
      if (myrow < P) uh[(long)it * P + myrow] = u;
    }
    __syncthreads();
    sk_col_partials(K, s_u, ti, tj, lane, s_part[wave]);
    __syncthreads();
    if (t < SKR) {
      float z = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) z += s_part[w][t];
      const float v = t < P ? 1.0f / z : 0.f;
      s_v[t] = v;
      if (t < P) vh[(long)(it + 1) * P + t] = v;
    }
    __syncthreads();
  }
}

// seeds of the reverse sweep (any P): G = gW .* K in place, gu_i = sum_j G_ij v_T,j -> dr_hist[niter-1], gv_j = sum_i G_ij u_T,i ->
// ds_hist[niter-1] (the sweep kernel reads them there before it overwrites those slots with dr / ds of the last iteration)
__global__ __launch_bounds__(SK_THREADS) void sinkhorn_seed_kernel(const float* __restrict__ Kin, const float* __restrict__ u_hist,
                                                                    const float* __restrict__ v_hist, float* __restrict__ gW,
                                                                    float* __restrict__ ds_hist, float* __restrict__ dr_hist, int P,
                                                                    int niter) {
  __shared__ float s_u[SK_MAXP], s_v[SK_MAXP], s_part[SK_THREADS];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* K = Kin + (long)b * P * P;
  float* G = gW + (long)b * P * P;
  for (long i = t; i < (long)P * P; i += SK_THREADS) G[i] *= K[i];
  for (int i = t; i < P; i += SK_THREADS) {
    s_v[i] = v_hist[((long)b * (niter + 1) + niter) * P + i];
    s_u[i] = u_hist[((long)b * niter + niter - 1) * P + i];
  }
  __threadfence_block();
  __syncthreads();
  row_matvec(G, s_v, dr_hist + ((long)b * niter + niter - 1) * P, P, wave, lane, SK_THREADS / 64);
  col_matvec(G, s_u, ds_hist + ((long)b * niter + niter - 1) * P, s_part, P, t);
}

__global__ __launch_bounds__(1024) void sinkhorn_bwd_reg_kernel(const float* __restrict__ Kin, const float* __restrict__ u_hist,
                                                                 const float* __restrict__ v_hist, float* __restrict__ ds_hist,
                                                                 float* __restrict__ dr_hist, int P, int niter) {
  __shared__ __attribute__((aligned(16))) float s_x[SKR], s_y[SKR], s_part[16][SKR];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, ti = t >> 5, tj = t & 31;
  const float* Kb = Kin + (long)b * P * P;
  const float* uh = u_hist + (long)b * niter * P;
  const float* vh = v_hist + (long)b * (niter + 1) * P;
  float* dsh = ds_hist + (long)b * niter * P;
  float* drh = dr_hist + (long)b * niter * P;
  SkTile K;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = 8 * ti + r;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int col = 8 * tj + c;
      K.k[r][c] = (row < P && col < P) ? Kb[row * P + col] : 0.f;
    }
  }
  const int myrow = sk_row_of(t, lane);
  for (int it = niter - 1; it >= 0; --it) {
    // gv (for v_{it+1}): the seed on the first pass, else the column sums of the previous pass; ds = -gv v^2
    if (t < SKR) {
      float z = 0.f;
      if (it == niter - 1) z = t < P ? dsh[(long)it * P + t] : 0.f;
      else {
#pragma unroll
        for (int w = 0; w < 16; ++w) z += s_part[w][t];
      }
      const float v = t < P ? vh[(long)(it + 1) * P + t] : 0.f;
      const float ds = -z * v * v;
      s_x[t] = ds;
      if (t < P) dsh[(long)it * P + t] = ds;
    }
    __syncthreads();
    float p[8];
    sk_row_partials(K, s_x, tj, p);
    const float y = sk_row_reduce(p, lane);
    if ((lane & 3) == 0) {
      const bool ok = myrow < P;
      const float u = ok ? uh[(long)it * P + myrow] : 0.f;
      const float seed = (it == niter - 1 && ok) ? drh[(long)it * P + myrow] : 0.f;
      const float dr = -(seed + y) * u * u;
      s_y[myrow] = dr;
      if (ok) drh[(long)it * P + myrow] = dr;
    }
    __syncthreads();
    sk_col_partials(K, s_y, ti, tj, lane, s_part[wave]);
    __syncthreads();
  }
}

// dS_ij += K_ij * ( G_ij / K_ij * u_T,i v_T,j + sum_t u_t,i ds_t,j + dr_t,i v_t,j )   (j != i), G = gW .* K from the sweep
// grid (P / 16 row blocks, images); thread j owns a column, 16 rows per block
__global__ __launch_bounds__(256) void sinkhorn_gk_kernel(const float* __restrict__ Kin, const float* __restrict__ G,
                                                          const float* __restrict__ u_hist, const float* __restrict__ v_hist,
                                                          const float* __restrict__ ds_hist, const float* __restrict__ dr_hist,
                                                          float* __restrict__ dS, int P, int niter) {
  extern __shared__ float s_rows[];  // [niter][16] u, then [niter][16] dr
  const int b = blockIdx.y, i0 = blockIdx.x * 16, t = threadIdx.x;
  const float* uh = u_hist + (long)b * niter * P;
  const float* vh = v_hist + (long)b * (niter + 1) * P;
  const float* dsh = ds_hist + (long)b * niter * P;
  const float* drh = dr_hist + (long)b * niter * P;
  float* s_u = s_rows;
  float* s_dr = s_rows + niter * 16;
  for (int e = t; e < niter * 16; e += 256) {
    const int it = e / 16, r = e % 16;
    const bool ok = i0 + r < P;
    s_u[e] = ok ? uh[(long)it * P + i0 + r] : 0.f;
    s_dr[e] = ok ? drh[(long)it * P + i0 + r] : 0.f;
  }
  __syncthreads();
  for (int j = t; j < P; j += 256) {
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int it = 0; it < niter; ++it) {
      const float ds = dsh[(long)it * P + j], v = vh[(long)it * P + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += s_u[it * 16 + r] * ds + s_dr[it * 16 + r] * v;
    }
    const float vT = vh[(long)niter * P + j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + r;
      if (i >= P || i == j) continue;
      const long o = ((long)b * P + i) * P + j;
      dS[o] += acc[r] * Kin[o] + G[o] * s_u[(niter - 1) * 16 + r] * vT;
    }
  }
}

// ---- cross-entropy over [l_pos | l_neg] / T (base_NCE.py:43-50,67-77; monce.py:24-31) ---------------------------------
// wave per patch row.  MONCE: l_neg_ij += T log(u_i exp(S_ij) v_j (num_patches - 1) + 1e-8) before the diagonal fill.
template <bool MONCE>
__global__ __launch_bounds__(256) void nce_ce_kernel(const float* __restrict__ S, const float* __restrict__ u, long ustride,
                                                     const float* __restrict__ v, long vstride, float* __restrict__ loss_rows, float* __restrict__ dS, float* __restrict__ gW,
                                                     long rows, int P, float T, float pm1, const float* __restrict__ grow, float* __restrict__ gpos,
                                                     float eps) {
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int i = row % P;
  const long b = row / P;
  const float* Sr = S + row * P;
  const float invT = 1.0f / T;
  const float pos = Sr[i] * invT;
  const float ui = MONCE ? u[b * ustride + i] : 0.f;
  const float* vb = MONCE ? v + b * vstride : nullptr;
  float mx = pos;
  for (int j = lane; j < P; j += 64) {
    float l;
    if (j == i) l = -10.0f * invT;
    else {
      l = Sr[j];
      if (MONCE) l += T * logf(ui * expf(Sr[j] / eps) * vb[j] * pm1 + 1e-8f);
      l *= invT;
    }
    mx = fmaxf(mx, l);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float se = 0.f;
  for (int j = lane; j < P; j += 64) {
    float l;
    if (j == i) l = -10.0f * invT;
    else {
      l = Sr[j];
      if (MONCE) l += T * logf(ui * expf(Sr[j] / eps) * vb[j] * pm1 + 1e-8f);
      l *= invT;
    }
    se += expf(l - mx);
  }
  se = wave_sum(se) + expf(pos - mx);
  const float lse = mx + logf(se);
  if (lane == 0) loss_rows[row] = lse - pos;
  if (!dS) return;
  const float gscale = grow[row];
  if (lane == 0) gpos[row] = gscale * (expf(pos - lse) - 1.0f) * invT;
  for (int j = lane; j < P; j += 64) {
    float g, gw = 0.f;
    if (j == i) g = 0.f;
    else {
      float l = Sr[j];
      float f = 0.f;
      if (MONCE) {
        f = ui * expf(Sr[j] / eps) * vb[j] * pm1 + 1e-8f;
        l += T * logf(f);
      }
      g = gscale * expf(l * invT - lse) * invT;
      if (MONCE) gw = g * T * pm1 / f;
    }
    dS[row * P + j] = g;
    if (MONCE) gW[row * P + j] = gw;
  }
}

// y[r][:] += g[r] * x[r][:]
__global__ void row_axpy_kernel(float* __restrict__ y, const float* __restrict__ g, const float* __restrict__ x, long R, int D) {
  const long total = R * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) y[i] += g[i / D] * x[i];
}

}  // namespace

extern "C" int jg_row_axpy(float* y, const float* g, const float* x, int64_t R, int D, jg_stream_t s) {
  if (!y || !g || !x || R < 1 || D < 1) return JG_ERR_BAD_ARG;
  const long total = (long)R * D;
  hipLaunchKernelGGL(row_axpy_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                     y, g, x, (long)R, D);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_sgemm(const float* A, const float* B, float* C, const float* bias, const float* E, int M, int N, int K, int64_t sam,
                        int64_t sak, int64_t sbn, int64_t sbk, int64_t scm, int64_t scn, int nbatch, int64_t ba, int64_t bb, int64_t bc,
                        float alpha, float beta, int act_a, int act_b, int act_e, jg_stream_t s) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || nbatch < 1 || nbatch > 65535) return JG_ERR_BAD_ARG;
  SgemmP p{A, B, C, bias, E, M, N, K, sam, sak, sbn, sbk, scm, scn, ba, bb, bc, alpha, beta, act_a, act_b, act_e, 1};
  // accumulating GEMMs with a long reduction and a small output (the nn.Linear weight gradient: 16 tiles, K = rows of the batch)
  // are split along K so that the launch fills the chip; partial tiles land with atomics
  const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  if (nbatch == 1 && beta == 1.0f && !bias && !E && tiles < 128 && K >= 256 && jg_tune(JG_TUNE_DETERMINISTIC) == 0) {     // (deterministic mode: no K split, no atomics)
    int ks = 256 / tiles;
    while (ks > 1 && K / ks < 64) ks >>= 1;
    p.ksplit = ks;
  }
  return sgemm_launch(p, nbatch, (hipStream_t)s);
}

extern "C" int jg_nce_sinkhorn_fwd(const float* S, float* K, float* u_hist, float* v_hist, int nimg, int P, int niter, float eps,
                                   jg_stream_t s) {
  if (!S || !K || !u_hist || !v_hist || nimg < 1 || P < 1 || niter < 1) return JG_ERR_BAD_ARG;
  if (P > SK_MAXP) return JG_ERR_UNSUPPORTED;
  if (P <= SKR && !jg_tune(JG_TUNE_SINKHORN_GENERIC))
    hipLaunchKernelGGL(sinkhorn_fwd_reg_kernel, dim3(nimg), dim3(1024), 0, (hipStream_t)s, S, K, u_hist, v_hist, P, niter, eps);
  else
    hipLaunchKernelGGL(sinkhorn_fwd_kernel, dim3(nimg), dim3(SK_THREADS), 0, (hipStream_t)s, S, K, u_hist, v_hist, P, niter, eps);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_nce_ce(const float* S, const float* u, int64_t ustride, const float* v, int64_t vstride, float* loss_rows, float* dS, float* gW, int nimg, int P, float T,
                         float pm1, const float* grow, float* gpos, float eps, jg_stream_t s) {
  if (!S || !loss_rows || nimg < 1 || P < 1 || T <= 0.f) return JG_ERR_BAD_ARG;
  const long rows = (long)nimg * P;
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (u) {
    if (!v || (dS && !gW)) return JG_ERR_BAD_ARG;
  }
  if (dS && (!grow || !gpos)) return JG_ERR_BAD_ARG;
  if (u) {
    hipLaunchKernelGGL((nce_ce_kernel<true>), grid, dim3(256), 0, (hipStream_t)s, S, u, (long)ustride, v, (long)vstride, loss_rows, dS, gW, rows, P, T, pm1, grow, gpos,
                       eps);
  } else {
    hipLaunchKernelGGL((nce_ce_kernel<false>), grid, dim3(256), 0, (hipStream_t)s, S, u, (long)ustride, v, (long)vstride, loss_rows, dS, gW, rows, P, T, pm1, grow, gpos,
                       eps);
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_nce_sinkhorn_bwd(const float* K, const float* u_hist, const float* v_hist, float* gW, float* ds_hist, float* dr_hist,
                                   float* dS, int nimg, int P, int niter, jg_stream_t s) {
  if (!K || !u_hist || !v_hist || !gW || !ds_hist || !dr_hist || !dS || nimg < 1 || P < 1 || niter < 1) return JG_ERR_BAD_ARG;
  if (P > SK_MAXP || niter * 32 * sizeof(float) > 48 * 1024) return JG_ERR_UNSUPPORTED;
  if (P <= SKR && !jg_tune(JG_TUNE_SINKHORN_GENERIC)) {
    hipLaunchKernelGGL(sinkhorn_seed_kernel, dim3(nimg), dim3(SK_THREADS), 0, (hipStream_t)s, K, u_hist, v_hist, gW, ds_hist, dr_hist, P, niter);
    hipLaunchKernelGGL(sinkhorn_bwd_reg_kernel, dim3(nimg), dim3(1024), 0, (hipStream_t)s, K, u_hist, v_hist, ds_hist, dr_hist, P, niter);
  } else
    hipLaunchKernelGGL(sinkhorn_bwd_kernel, dim3(nimg), dim3(SK_THREADS), 0, (hipStream_t)s, K, u_hist, v_hist, gW, ds_hist, dr_hist, P,
                       niter);
  hipLaunchKernelGGL(sinkhorn_gk_kernel, dim3((P + 15) / 16, nimg), dim3(256), niter * 32 * sizeof(float), (hipStream_t)s, K, gW, u_hist,
                     v_hist, ds_hist, dr_hist, dS, P, niter);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

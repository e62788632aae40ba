// Kernels of the ViT feature network of the projected discriminator (D_proj_network_type "vitsmall": timm vit_small_patch16_224,
// /root/reference/models/modules/projected_d/projector.py:138-153,327-331) on gfx950.  The linear layers run on the MFMA GEMM of
// gemm_nt.hip and LayerNorm forward on segformer.hip; here:
//
//   vit_attn_*        multi-head self-attention over token sequences of ANY length (257 = 16 x 16 patches + class token) with head
//                     dim 32 or 64, q / k / v as three strided views of one projection ([B, T, 3C], channel = which * C + head * D + d);
//                     same two-pass structure as attention.hip (logits never leave the chip, S^T = K Q^T so that the P operand of the
//                     second MFMA needs no shuffle), a head dim of 64 = two 32-wide halves of the same LDS images; keys beyond T are
//                     masked, rows beyond T are loaded as zeros and never stored                                   MFMA / v_exp
//   vit_tokens_*      [class token | patch embeddings] + position embedding; its adjoint                            HBM
//   gelu_*            exact (erf) GELU and its derivative                                                            HBM
//   ln_bwd_res        LayerNorm input gradient of a FROZEN layer plus the residual branch: dx = res + LN'(dy)         HBM
//   unpatchify        adjoint of the non-overlapping patch gather of the 16 x 16 / stride 16 embedding convolution    HBM
#include "common.h"

namespace {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_v4;

constexpr int BT = 64;       // rows per block / per streamed chunk

// [row][32] 16-bit tiles in LDS (64-byte rows = 4 chunks of 16 B): chunk XOR for the ds_read_b128 operand fetch, 32-byte-block XOR
// for the transposing reads (same images as attention.hip)
__device__ __forceinline__ int fsw(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int bsw(int row) { return (row >> 2) & 1; }

// one 16-byte chunk per thread of a [64][32] tile whose first row is `row0` of a [nrows][ld] matrix; rows >= nrows read as zeros
template <typename T>
__device__ __forceinline__ uint4 stage_load(const T* __restrict__ g, long ld, int row0, int nrows, int tid) {
  const int row = tid >> 2, c = tid & 3;
  const bool ok = row0 + row < nrows;
  return ldg16(g + (ok ? (long)(row0 + row) * ld + c * 8 : 0), ok);
}
template <int MODE>
__device__ __forceinline__ void stage_store(uint4* lds, int tid, const uint4& r) {
  const int row = tid >> 2, c = tid & 3;
  const int pc = MODE == 0 ? (c ^ fsw(row)) : ((((c >> 1) ^ bsw(row)) << 1) | (c & 1));
  lds[row * 4 + pc] = r;
}
__device__ __forceinline__ uint4 frag_rows(const uint4* lds, int tile, int l15, int g) {
  const int row = tile * 16 + l15;
  return lds[row * 4 + (g ^ fsw(row))];
}
// lane (col dt * 16 + l15, k-group g) receives rows {blk*32 + g*4 + 0..3, blk*32 + 16 + g*4 + 0..3}
__device__ __forceinline__ uint4 frag_tr(const uint4* lds, int blk, int dt, int l15, int g) {
  const char* base = reinterpret_cast<const char*>(lds);
  uint32_t w[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = blk * 32 + h * 16 + g * 4 + (l15 >> 2);
    const int off = row * 64 + ((dt ^ bsw(row)) << 5) + (l15 & 3) * 8;
    const uint2 u = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + off)));
    w[2 * h] = u.x;
    w[2 * h + 1] = u.y;
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}
template <typename T> __device__ __forceinline__ uint4 pack_p(const f32x4& a, const f32x4& b) {
  float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return pack8<T>(f);
}

struct VitAttnP {
  const char *q, *k, *v;      // three views of the projection output, row stride ld, head h at + h * hs (elements)
  long ld, hs;
  char* o;                    // [B, T, nh * D], head h at h * D
  float* L;                   // [B * nh, T] logsumexp
  const char* dO;             // [B, T, nh * D]
  float* Dq;                  // [B * nh, T] rowsum(dO * O)
  char *dq, *dk, *dv;         // views of the projection gradient, row stride ldd, head stride hs
  long ldd;
  int B, T, nh;
  float scale;
};

// ------------------------------------------------------------------------------------------------------------
// forward: grid (ceil(T / 64), B * heads), 4 waves x 16 queries
// ------------------------------------------------------------------------------------------------------------
template <typename T, int NH>
__global__ __launch_bounds__(256) void vit_attn_fwd_kernel(VitAttnP p) {
  constexpr int D = 32 * NH;
  __shared__ uint4 sK[NH][BT * 4], sV[NH][BT * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const long boff = (long)b * p.T * p.ld + (long)h * p.hs;
  const T *qb = (const T*)p.q + boff, *kb = (const T*)p.k + boff, *vb = (const T*)p.v + boff;
  const int q0 = blockIdx.x * BT + wave * 16;
  const int qrow = min(q0 + l15, p.T - 1);

  uint4 fq[NH];               // lane (query l15, k-group g): q[query][hf * 32 + g * 8 .. + 7]
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) fq[hf] = *reinterpret_cast<const uint4*>(qb + (long)qrow * p.ld + hf * 32 + g * 8);

  const int nchunk = (p.T + BT - 1) / BT;
  uint4 rk[NH], rv[NH];

  // ---- pass 1: logsumexp per query (per-lane online partials over the lane's own keys) ----
  float mrun = -1e30f, lrun = 0.f;
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) rk[hf] = stage_load<T>(kb + hf * 32, p.ld, c * BT, p.T, tid);
    __syncthreads();
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) stage_store<0>(sK[hf], tid, rk[hf]);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) s = Mfma<T>::run(frag_rows(sK[hf], kt, l15, g), fq[hf], s);
      const int key0 = c * BT + kt * 16 + g * 4;
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, key0 + r < p.T ? s[r] * p.scale : -1e30f);
      const float mnew = fmaxf(mrun, mx);
      float acc = lrun * __expf(mrun - mnew);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc += key0 + r < p.T ? __expf(s[r] * p.scale - mnew) : 0.f;
      mrun = mnew;
      lrun = acc;
    }
  }
  float Lq;
  {
    float m = mrun;
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = lrun * __expf(mrun - m);
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    Lq = m + __logf(l);
    if (g == 0 && q0 + l15 < p.T) p.L[(long)bh * p.T + q0 + l15] = Lq;
  }

  // ---- pass 2: O = sum_keys exp(S - L) V ----
  f32x4 o[NH * 2];
#pragma unroll
  for (int i = 0; i < NH * 2; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      rk[hf] = stage_load<T>(kb + hf * 32, p.ld, c * BT, p.T, tid);
      rv[hf] = stage_load<T>(vb + hf * 32, p.ld, c * BT, p.T, tid);
    }
    __syncthreads();
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      stage_store<0>(sK[hf], tid, rk[hf]);
      stage_store<1>(sV[hf], tid, rv[hf]);
    }
    __syncthreads();
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {   // 32 keys = two 16-key tiles = one P operand
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) {
        s0 = Mfma<T>::run(frag_rows(sK[hf], 2 * kb2, l15, g), fq[hf], s0);
        s1 = Mfma<T>::run(frag_rows(sK[hf], 2 * kb2 + 1, l15, g), fq[hf], s1);
      }
      const int key0 = c * BT + kb2 * 32 + g * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = key0 + r < p.T ? __expf(s0[r] * p.scale - Lq) : 0.f;
        s1[r] = key0 + 16 + r < p.T ? __expf(s1[r] * p.scale - Lq) : 0.f;
      }
      const uint4 fp = pack_p<T>(s0, s1);
#pragma unroll
      for (int hf = 0; hf < NH; ++hf)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[hf * 2 + dt] = Mfma<T>::run(fp, frag_tr(sV[hf], kb2, dt, l15, g), o[hf * 2 + dt]);
    }
  }
  // O tile: row = query g*4 + r, col = d = i*16 + l15
  T* ob = (T*)p.o + (long)b * p.T * (p.nh * D) + h * D;
#pragma unroll
  for (int i = 0; i < NH * 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + g * 4 + r;
      if (q < p.T) ob[(long)q * (p.nh * D) + i * 16 + l15] = from_f32<T>(o[i][r]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward 1: dQ (and Dq = rowsum(dO * O)); a block owns 64 queries and streams the keys
// ------------------------------------------------------------------------------------------------------------
template <typename T, int NH>
__global__ __launch_bounds__(256) void vit_attn_bwd_dq_kernel(VitAttnP p) {
  constexpr int D = 32 * NH;
  __shared__ uint4 sK[NH][BT * 4], sKt[NH][BT * 4], sV[NH][BT * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const long boff = (long)b * p.T * p.ld + (long)h * p.hs;
  const int C1 = p.nh * D;
  const T *qb = (const T*)p.q + boff, *kb = (const T*)p.k + boff, *vb = (const T*)p.v + boff;
  const T* dab = (const T*)p.dO + (long)b * p.T * C1 + h * D;
  const T* ob = (const T*)p.o + (long)b * p.T * C1 + h * D;
  const int q0 = blockIdx.x * BT + wave * 16;
  const int qrow = min(q0 + l15, p.T - 1);

  uint4 fq[NH], fdo[NH];
  float Dq = 0.f;
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
    fq[hf] = *reinterpret_cast<const uint4*>(qb + (long)qrow * p.ld + hf * 32 + g * 8);
    fdo[hf] = *reinterpret_cast<const uint4*>(dab + (long)qrow * C1 + hf * 32 + g * 8);
    const uint4 fo = *reinterpret_cast<const uint4*>(ob + (long)qrow * C1 + hf * 32 + g * 8);
    float a[8], c[8];
    unpack8<T>(fdo[hf], a);
    unpack8<T>(fo, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) Dq += a[i] * c[i];
  }
  Dq += __shfl_xor(Dq, 16);
  Dq += __shfl_xor(Dq, 32);
  const float Lq = p.L[(long)bh * p.T + qrow];
  if (g == 0 && q0 + l15 < p.T) p.Dq[(long)bh * p.T + q0 + l15] = Dq;

  f32x4 dq[NH * 2];
#pragma unroll
  for (int i = 0; i < NH * 2; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nchunk = (p.T + BT - 1) / BT;
  uint4 rk[NH], rv[NH];
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      rk[hf] = stage_load<T>(kb + hf * 32, p.ld, c * BT, p.T, tid);
      rv[hf] = stage_load<T>(vb + hf * 32, p.ld, c * BT, p.T, tid);
    }
    __syncthreads();
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      stage_store<0>(sK[hf], tid, rk[hf]);
      stage_store<1>(sKt[hf], tid, rk[hf]);
      stage_store<0>(sV[hf], tid, rv[hf]);
    }
    __syncthreads();
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) {
        s0 = Mfma<T>::run(frag_rows(sK[hf], 2 * kb2, l15, g), fq[hf], s0);          // S^T  [key][q]
        s1 = Mfma<T>::run(frag_rows(sK[hf], 2 * kb2 + 1, l15, g), fq[hf], s1);
        d0 = Mfma<T>::run(frag_rows(sV[hf], 2 * kb2, l15, g), fdo[hf], d0);         // dP^T [key][q]
        d1 = Mfma<T>::run(frag_rows(sV[hf], 2 * kb2 + 1, l15, g), fdo[hf], d1);
      }
      const int key0 = c * BT + kb2 * 32 + g * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = key0 + r < p.T ? __expf(s0[r] * p.scale - Lq) * (d0[r] - Dq) : 0.f;
        s1[r] = key0 + 16 + r < p.T ? __expf(s1[r] * p.scale - Lq) * (d1[r] - Dq) : 0.f;
      }
      const uint4 fds = pack_p<T>(s0, s1);
#pragma unroll
      for (int hf = 0; hf < NH; ++hf)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dq[hf * 2 + dt] = Mfma<T>::run(fds, frag_tr(sKt[hf], kb2, dt, l15, g), dq[hf * 2 + dt]);
    }
  }
  T* dqb = (T*)p.dq + (long)b * p.T * p.ldd + (long)h * p.hs;
#pragma unroll
  for (int i = 0; i < NH * 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + g * 4 + r;
      if (q < p.T) dqb[(long)q * p.ldd + i * 16 + l15] = from_f32<T>(dq[i][r] * p.scale);
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward 2: dK, dV; a block owns 64 keys (a wave 16) and streams the queries
// ------------------------------------------------------------------------------------------------------------
template <typename T, int NH>
__global__ __launch_bounds__(256) void vit_attn_bwd_dkv_kernel(VitAttnP p) {
  constexpr int D = 32 * NH;
  __shared__ uint4 sQ[NH][BT * 4], sQt[NH][BT * 4], sO[NH][BT * 4], sOt[NH][BT * 4];
  __shared__ float sL[BT], sD[BT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.nh, h = bh % p.nh;
  const long boff = (long)b * p.T * p.ld + (long)h * p.hs;
  const int C1 = p.nh * D;
  const T *qb = (const T*)p.q + boff, *kb = (const T*)p.k + boff, *vb = (const T*)p.v + boff;
  const T* dab = (const T*)p.dO + (long)b * p.T * C1 + h * D;
  const int k0 = blockIdx.x * BT + wave * 16;
  const int krow = min(k0 + l15, p.T - 1);

  uint4 fk[NH], fv[NH];       // lane (key l15, k-group g)
#pragma unroll
  for (int hf = 0; hf < NH; ++hf) {
    fk[hf] = *reinterpret_cast<const uint4*>(kb + (long)krow * p.ld + hf * 32 + g * 8);
    fv[hf] = *reinterpret_cast<const uint4*>(vb + (long)krow * p.ld + hf * 32 + g * 8);
  }
  f32x4 dk[NH * 2], dv[NH * 2];
#pragma unroll
  for (int i = 0; i < NH * 2; ++i) dk[i] = dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunk = (p.T + BT - 1) / BT;
  uint4 rq[NH], ro[NH];
  for (int c = 0; c < nchunk; ++c) {
    float rl = 1e30f, rd = 0.f;          // queries beyond T: P = exp(S - 1e30) = 0
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      rq[hf] = stage_load<T>(qb + hf * 32, p.ld, c * BT, p.T, tid);
      ro[hf] = stage_load<T>(dab + hf * 32, C1, c * BT, p.T, tid);
    }
    if (tid < BT && c * BT + tid < p.T) {
      rl = p.L[(long)bh * p.T + c * BT + tid];
      rd = p.Dq[(long)bh * p.T + c * BT + tid];
    }
    __syncthreads();
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      stage_store<0>(sQ[hf], tid, rq[hf]);
      stage_store<1>(sQt[hf], tid, rq[hf]);
      stage_store<0>(sO[hf], tid, ro[hf]);
      stage_store<1>(sOt[hf], tid, ro[hf]);
    }
    if (tid < BT) {
      sL[tid] = rl;
      sD[tid] = rd;
    }
    __syncthreads();
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {   // 32 queries = two 16-query tiles = one P^T / dS^T operand
      const float4 L0 = *reinterpret_cast<const float4*>(&sL[qb2 * 32 + g * 4]);
      const float4 L1 = *reinterpret_cast<const float4*>(&sL[qb2 * 32 + 16 + g * 4]);
      const float4 D0 = *reinterpret_cast<const float4*>(&sD[qb2 * 32 + g * 4]);
      const float4 D1 = *reinterpret_cast<const float4*>(&sD[qb2 * 32 + 16 + g * 4]);
      const float l0[4] = {L0.x, L0.y, L0.z, L0.w}, l1[4] = {L1.x, L1.y, L1.z, L1.w};
      const float e0[4] = {D0.x, D0.y, D0.z, D0.w}, e1[4] = {D1.x, D1.y, D1.z, D1.w};
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
      for (int hf = 0; hf < NH; ++hf) {
        s0 = Mfma<T>::run(frag_rows(sQ[hf], 2 * qb2, l15, g), fk[hf], s0);          // S  [q][key]
        s1 = Mfma<T>::run(frag_rows(sQ[hf], 2 * qb2 + 1, l15, g), fk[hf], s1);
        d0 = Mfma<T>::run(frag_rows(sO[hf], 2 * qb2, l15, g), fv[hf], d0);          // dP [q][key]
        d1 = Mfma<T>::run(frag_rows(sO[hf], 2 * qb2 + 1, l15, g), fv[hf], d1);
      }
      f32x4 ds0, ds1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = __expf(s0[r] * p.scale - l0[r]);
        s1[r] = __expf(s1[r] * p.scale - l1[r]);
        ds0[r] = s0[r] * (d0[r] - e0[r]);
        ds1[r] = s1[r] * (d1[r] - e1[r]);
      }
      const uint4 fp = pack_p<T>(s0, s1), fds = pack_p<T>(ds0, ds1);
#pragma unroll
      for (int hf = 0; hf < NH; ++hf)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[hf * 2 + dt] = Mfma<T>::run(fp, frag_tr(sOt[hf], qb2, dt, l15, g), dv[hf * 2 + dt]);
          dk[hf * 2 + dt] = Mfma<T>::run(fds, frag_tr(sQt[hf], qb2, dt, l15, g), dk[hf * 2 + dt]);
        }
    }
  }
  T* dkb = (T*)p.dk + (long)b * p.T * p.ldd + (long)h * p.hs;
  T* dvb = (T*)p.dv + (long)b * p.T * p.ldd + (long)h * p.hs;
#pragma unroll
  for (int i = 0; i < NH * 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = k0 + g * 4 + r;
      if (key < p.T) {
        dkb[(long)key * p.ldd + i * 16 + l15] = from_f32<T>(dk[i][r] * p.scale);
        dvb[(long)key * p.ldd + i * 16 + l15] = from_f32<T>(dv[i][r]);
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
// element-wise helpers: one 16-byte chunk (8 channels) per thread
// ------------------------------------------------------------------------------------------------------------
// y[b, 0] = cls + pos[0];  y[b, 1 + n] = patch[b, n] + pos[1 + n]
template <typename T>
__global__ void vit_tokens_fwd_kernel(const T* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos, T* __restrict__ y,
                                      int B, int N, int C) {
  const int oc = C / 8;
  const long total = (long)B * (N + 1) * oc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % oc) * 8;
    const long bt = i / oc;
    const int t = (int)(bt % (N + 1));
    const long b = bt / (N + 1);
    float f[8];
    if (t == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = cls[c + j];
    } else {
      unpack8<T>(*reinterpret_cast<const uint4*>(patch + (b * N + t - 1) * C + c), f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] += pos[(long)t * C + c + j];
    *reinterpret_cast<uint4*>(y + bt * C + c) = pack8<T>(f);
  }
}
template <typename T>
__global__ void vit_tokens_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dpatch, int B, int N, int C) {
  const int oc = C / 8;
  const long total = (long)B * N * oc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % oc) * 8;
    const long bn = i / oc;
    const long b = bn / N, n = bn % N;
    *reinterpret_cast<uint4*>(dpatch + bn * C + c) = *reinterpret_cast<const uint4*>(dy + (b * (N + 1) + n + 1) * C + c);
  }
}

__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float u) {
  return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8<T>(reinterpret_cast<const uint4*>(x)[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = gelu_f(f[j]);
    reinterpret_cast<uint4*>(y)[i] = pack8<T>(f);
  }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8], d[8];
    unpack8<T>(reinterpret_cast<const uint4*>(x)[i], f);
    unpack8<T>(reinterpret_cast<const uint4*>(dy)[i], d);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = d[j] * gelu_grad_f(f[j]);
    reinterpret_cast<uint4*>(dx)[i] = pack8<T>(f);
  }
}

// dx = res + rstd (g - mean(g) - xh mean(g xh)),  g = dy gamma,  xh = (x - mean) rstd: one wave per row, C <= 512
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_res_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                         const float* __restrict__ mr, const T* __restrict__ res, T* __restrict__ dx, long R, int C) {
  const int lane = threadIdx.x & 63;
  const bool act = lane * 8 < C;
  const int cc = act ? lane * 8 : 0;
  float g[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] = act ? gamma[cc + j] : 0.f;
  const float inv_c = 1.0f / (float)C;
  for (long row = blockIdx.x * 4L + (threadIdx.x >> 6); row < R; row += gridDim.x * 4L) {
    const uint4 vx = *reinterpret_cast<const uint4*>(x + row * C + cc);
    const uint4 vd = *reinterpret_cast<const uint4*>(dy + row * C + cc);
    uint4 vr = make_uint4(0u, 0u, 0u, 0u);
    if (res) vr = *reinterpret_cast<const uint4*>(res + row * C + cc);
    const float mean = mr[row * 2], rstd = mr[row * 2 + 1];
    float f[8], d[8], rr[8], xh[8], gy[8];
    unpack8<T>(vx, f);
    unpack8<T>(vd, d);
    unpack8<T>(vr, rr);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xh[j] = (f[j] - mean) * rstd;
      gy[j] = d[j] * g[j];
      s1 += gy[j];
      s2 += gy[j] * xh[j];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (act) {
      const float m1 = s1 * inv_c, m2 = s2 * inv_c;
      float o8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] = rr[j] + rstd * (gy[j] - m1 - xh[j] * m2);
      *reinterpret_cast<uint4*>(dx + row * C + cc) = pack8<T>(o8);
    }
  }
}

// dimg[b, ph*P + r, pw*P + s, ci] = dpatch[(b, ph, pw)][ci * P*P + (P-1-r) * P + (P-1-s)]   (the column order of the flipped / transposed
// working weights w16T [Cin][R][S][Cout] that the input-gradient GEMM multiplies with); 8 channels per pixel
template <typename T>
__global__ void unpatchify_kernel(const T* __restrict__ dpatch, T* __restrict__ dimg, int B, int Hp, int Wp, int P) {
  const int H = Hp * P, W = Wp * P;
  const long total = (long)B * H * W;
  const long ldp = 8L * P * P;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long by = i / W;
    const int y = (int)(by % H);
    const long b = by / H;
    const int r = y % P, s = x % P;
    const T* src = dpatch + ((b * Hp + y / P) * Wp + x / P) * ldp + (P - 1 - r) * P + (P - 1 - s);
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = (uint32_t)to_bits<T>(src[(long)(2 * j) * P * P]) | ((uint32_t)to_bits<T>(src[(long)(2 * j + 1) * P * P]) << 16);
    *reinterpret_cast<uint4*>(dimg + i * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// dst[b][c][r] = src[b][r][c] (2-byte elements, any R / Cc): 64 x 64 tiles through LDS, reads and writes both contiguous
template <typename T>
__global__ __launch_bounds__(256) void transpose2d_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int Cc) {
  __shared__ uint16_t tile[64][65];
  const long b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const uint16_t* s = reinterpret_cast<const uint16_t*>(src) + b * (long)R * Cc;
  uint16_t* d = reinterpret_cast<uint16_t*>(dst) + b * (long)R * Cc;
  for (int j = ty; j < 64; j += 4)
    if (r0 + j < R && c0 + tx < Cc) tile[j][tx] = s[(long)(r0 + j) * Cc + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 64; j += 4)
    if (c0 + j < Cc && r0 + tx < R) d[(long)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

inline unsigned grid_for8(long n, long cap = 16384) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define JG_VIT_DISPATCH_NH(hd, ...)                          \
  do {                                                       \
    if ((hd) == 32) { constexpr int NH = 1; __VA_ARGS__ }    \
    else { constexpr int NH = 2; __VA_ARGS__ }               \
  } while (0)

extern "C" int jg_vit_attention_fwd(int dtype, const void* q, const void* k, const void* v, int64_t ld, int64_t head_stride, void* o, float* L,
                                    int B, int T, int heads, int head_dim, float scale, jg_stream_t s) {
  if (!q || !k || !v || !o || !L || B < 1 || T < 1 || heads < 1 || ld % 8 || head_stride % 8) return JG_ERR_BAD_ARG;
  if (head_dim != 32 && head_dim != 64) return JG_ERR_UNSUPPORTED;
  VitAttnP p = {};
  p.q = (const char*)q; p.k = (const char*)k; p.v = (const char*)v; p.ld = ld; p.hs = head_stride; p.o = (char*)o; p.L = L;
  p.B = B; p.T = T; p.nh = heads; p.scale = scale;
  dim3 grid((T + BT - 1) / BT, B * heads);
  JG_DISPATCH_DTYPE(dtype, JG_VIT_DISPATCH_NH(head_dim, hipLaunchKernelGGL((vit_attn_fwd_kernel<T, NH>), grid, dim3(256), 0, (hipStream_t)s, p);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_vit_attention_bwd(int dtype, const void* q, const void* k, const void* v, int64_t ld, int64_t head_stride, const void* o,
                                    const float* L, const void* d_o, void* dq, void* dk, void* dv, int64_t ldd, float* Dq, int B, int T, int heads,
                                    int head_dim, float scale, jg_stream_t s) {
  if (!q || !k || !v || !o || !L || !d_o || !dq || !dk || !dv || !Dq || B < 1 || T < 1 || heads < 1 || ld % 8 || ldd % 8 || head_stride % 8)
    return JG_ERR_BAD_ARG;
  if (head_dim != 32 && head_dim != 64) return JG_ERR_UNSUPPORTED;
  VitAttnP p = {};
  p.q = (const char*)q; p.k = (const char*)k; p.v = (const char*)v; p.ld = ld; p.hs = head_stride; p.o = (char*)const_cast<void*>(o);
  p.L = const_cast<float*>(L); p.dO = (const char*)d_o; p.dq = (char*)dq; p.dk = (char*)dk; p.dv = (char*)dv; p.ldd = ldd; p.Dq = Dq;
  p.B = B; p.T = T; p.nh = heads; p.scale = scale;
  dim3 grid((T + BT - 1) / BT, B * heads);
  JG_DISPATCH_DTYPE(dtype, JG_VIT_DISPATCH_NH(head_dim, hipLaunchKernelGGL((vit_attn_bwd_dq_kernel<T, NH>), grid, dim3(256), 0, (hipStream_t)s, p);
                                              hipLaunchKernelGGL((vit_attn_bwd_dkv_kernel<T, NH>), grid, dim3(256), 0, (hipStream_t)s, p);););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_vit_tokens_fwd(int dtype, const void* patch, const float* cls, const float* pos, void* y, int B, int N, int C, jg_stream_t s) {
  if (!patch || !cls || !pos || !y || B < 1 || N < 1 || C < 8 || C % 8) return JG_ERR_BAD_ARG;
  const long total = (long)B * (N + 1) * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((vit_tokens_fwd_kernel<T>), dim3(grid_for8(total)), dim3(256), 0, (hipStream_t)s, (const T*)patch, cls,
                                              pos, (T*)y, B, N, C););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_vit_tokens_bwd(int dtype, const void* dy, void* dpatch, int B, int N, int C, jg_stream_t s) {
  if (!dy || !dpatch || B < 1 || N < 1 || C < 8 || C % 8) return JG_ERR_BAD_ARG;
  const long total = (long)B * N * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((vit_tokens_bwd_kernel<T>), dim3(grid_for8(total)), dim3(256), 0, (hipStream_t)s, (const T*)dy,
                                              (T*)dpatch, B, N, C););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_gelu_fwd(int dtype, const void* x, void* y, int64_t n, jg_stream_t s) {
  if (!x || !y || n < 8 || n % 8) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gelu_fwd_kernel<T>), dim3(grid_for8(n / 8)), dim3(256), 0, (hipStream_t)s, (const T*)x, (T*)y, (long)(n / 8)););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_gelu_bwd(int dtype, const void* x, const void* dy, void* dx, int64_t n, jg_stream_t s) {
  if (!x || !dy || !dx || n < 8 || n % 8) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(grid_for8(n / 8)), dim3(256), 0, (hipStream_t)s, (const T*)x, (const T*)dy,
                                              (T*)dx, (long)(n / 8)););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_layernorm_bwd_res(int dtype, const void* x, const void* dy, const float* gamma, const float* mr, const void* res, void* dx,
                                    int64_t R, int C, jg_stream_t s) {
  if (!x || !dy || !gamma || !mr || !dx || R < 1 || C < 8 || C % 8 || C > 512) return JG_ERR_BAD_ARG;
  const long blocks = (R + 3) / 4;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_res_kernel<T>), dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)x, (const T*)dy, gamma, mr, (const T*)res, (T*)dx, (long)R, C););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_unpatchify(int dtype, const void* dpatch, void* dimg, int B, int Hp, int Wp, int P, jg_stream_t s) {
  if (!dpatch || !dimg || B < 1 || Hp < 1 || Wp < 1 || P < 1) return JG_ERR_BAD_ARG;
  const long total = (long)B * Hp * P * Wp * P;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((unpatchify_kernel<T>), dim3(grid_for8(total)), dim3(256), 0, (hipStream_t)s, (const T*)dpatch,
                                              (T*)dimg, B, Hp, Wp, P););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_transpose2d(int dtype, const void* src, void* dst, int B, int R, int C, jg_stream_t s) {
  if (!src || !dst || B < 1 || B > 65535 || R < 1 || C < 1) return JG_ERR_BAD_ARG;
  dim3 grid((C + 63) / 64, (R + 63) / 64, B);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose2d_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (const T*)src, (T*)dst, R, C););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

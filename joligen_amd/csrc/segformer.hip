// Kernels of the SegFormer attention generator (models/modules/segformer/*.py, attn_network.py), NHWC 16-bit
// activations (a token sequence [B, N, C] IS the NHWC map [B, H, W, C]), fp32 parameters and statistics:
//   layernorm fwd/bwd        nn.LayerNorm(C, eps 1e-6) over the channel row of every token          HBM-bound, one pass each
//   dwconv3x3(+GELU) fwd/bwd the positional depth-wise conv of MixFFN with the following nn.GELU()    HBM-bound
//   attn_smallkv fwd/bwd     nn.MultiheadAttention core with the spatially-reduced key/value set (T_kv <= 256, head dim 32)
//   bilinear fwd/bwd         F.interpolate(mode="bilinear", align_corners=False) written straight into the concat buffer
//   bn_coef / bn_bwd_coef    nn.BatchNorm2d on top of the (b, c) sums of the GroupNorm kernels (ResnetDecoder tail)
//   attn_compose fwd/bwd     10-way softmax attention x (9 generated images | input) blend of BaseGenerator_attn
//   scale_rows / scale_bc    DropPath (per sample) and Dropout2d (per sample and channel) as multiplications by given factors
#include "common.h"

namespace {

inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// nn.GELU() (erf form) and its derivative Phi(x) + x phi(x).  The depth-wise kernels are bound by their vector instructions (8 channels
// x 9 taps of fp32 FMAs per 16-byte chunk); with the library erff / expf, which branch per element, and a 64-bit p % W, (p / W) % H per
// step a pixel chunk of the weight gradient was ~850 instructions, ~110 of them useful.  Here erf is Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, well below the 16-bit rounding of the result), whose e^{-z^2} at z = x / sqrt 2 is the e^{-x^2/2} of the density
// term: ONE v_exp_f32 and one v_rcp_f32 per element, no branches.  (Round 3 tried the same formula while the kernels still spent their
// time in 1024-deep atomic chains and 64-bit divisions and saw nothing; after those went -- DwWalk, dw_partials_sum_kernel -- it pays.)
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);          // e^{-x^2/2}
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erf_abs = fmaf(-p * t, e, 1.0f);                                      // erf(|x| / sqrt 2)
  return fmaf(x * 0.3989422804014327f, e, fmaf(0.5f, copysignf(erf_abs, x), 0.5f));
}
__device__ __forceinline__ float gelu_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  return x * fmaf(0.5f, copysignf(fmaf(-p * t, e, 1.0f), x), 0.5f);
}
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
// two 16-bit values of one dword -> fp32 pair (bf16: a shift and a mask)
template <typename T> __device__ __forceinline__ dw_f32x2 dw_unpack2(uint32_t w) {
  dw_f32x2 r;
  if constexpr (sizeof(T) == 2 && __is_same(T, bf16_t)) {
    r.x = __uint_as_float(w << 16);
    r.y = __uint_as_float(w & 0xffff0000u);
  } else {
    r.x = to_f32(bits_to<T>((uint16_t)(w & 0xffffu)));
    r.y = to_f32(bits_to<T>((uint16_t)(w >> 16)));
  }
  return r;
}

// the nine 16-byte neighbour loads of a 3x3 window issued back to back with ONE wait, as a single asm statement (clang puts an ordinary
// load next to its first use, and the old per-tap `continue` on the image border made every tap its own L2 round trip: a 17 MB launch took
// 65 us).  Out-of-image taps read a clamped (valid) address and are zeroed afterwards.
typedef unsigned dw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dw_load9(uint4 (&v)[9], const void* const (&p)[9]) {
  dw_u32x4 r[9];
  asm volatile("global_load_dwordx4 %0, %9, off\n\tglobal_load_dwordx4 %1, %10, off\n\tglobal_load_dwordx4 %2, %11, off\n\t"
               "global_load_dwordx4 %3, %12, off\n\tglobal_load_dwordx4 %4, %13, off\n\tglobal_load_dwordx4 %5, %14, off\n\t"
               "global_load_dwordx4 %6, %15, off\n\tglobal_load_dwordx4 %7, %16, off\n\tglobal_load_dwordx4 %8, %17, off\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8])
               : "memory");
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = make_uint4(r[i].x, r[i].y, r[i].z, r[i].w);
}
// the same with two more rows in front (the backward's dy and pre-activation of the centre pixel)
__device__ __forceinline__ void dw_load11(uint4 (&v)[11], const void* const (&p)[11]) {
  dw_u32x4 r[11];
  asm volatile("global_load_dwordx4 %0, %11, off\n\tglobal_load_dwordx4 %1, %12, off\n\tglobal_load_dwordx4 %2, %13, off\n\t"
               "global_load_dwordx4 %3, %14, off\n\tglobal_load_dwordx4 %4, %15, off\n\tglobal_load_dwordx4 %5, %16, off\n\t"
               "global_load_dwordx4 %6, %17, off\n\tglobal_load_dwordx4 %7, %18, off\n\tglobal_load_dwordx4 %8, %19, off\n\t"
               "global_load_dwordx4 %9, %20, off\n\tglobal_load_dwordx4 %10, %21, off\n\ts_waitcnt vmcnt(0)"
               : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]),
                 "=&v"(r[10])
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10])
               : "memory");
#pragma unroll
  for (int i = 0; i < 11; ++i) v[i] = make_uint4(r[i].x, r[i].y, r[i].z, r[i].w);
}
// (px, py) of a pixel index that advances by a fixed stride: one 32-bit division pair up front, then adds and two wrap-arounds per step
// (the loops used to compute p % W and (p / W) % H on a 64-bit p every iteration: ~400 of the ~850 instructions of a pixel chunk)
struct DwWalk {
  int px, py, sx, sy, W, H;
  __device__ __forceinline__ DwWalk(unsigned p0, unsigned stride, int W_, int H_) : W(W_), H(H_) {
    px = (int)(p0 % (unsigned)W_);
    py = (int)((p0 / (unsigned)W_) % (unsigned)H_);
    sx = (int)(stride % (unsigned)W_);
    sy = (int)((stride / (unsigned)W_) % (unsigned)H_);
  }
  __device__ __forceinline__ void step() {
    px += sx;
    py += sy;
    if (px >= W) {
      px -= W;
      py += 1;
    }
    if (py >= H) py -= H;
  }
};
// window of pixel (py, px) of an [H, W, C] image row-major at `base` (pointer to the centre pixel's chunk): pointers + validity of the 9 taps
// reflect (round 6: padding_mode = 'reflect' of the depth-wise convolutions of the mobile ResNet blocks, mobile_modules.py:4-40): an
// out-of-image tap reads the mirrored pixel (-1 -> 1, H -> H - 2) instead of a zero -- no padded copy of x, no crop of the result
template <typename T, bool FLIP>
__device__ __forceinline__ void dw_window(const T* centre, int px, int py, int H, int W, int C, const void* (&ptr)[9], bool (&ok)[9], int reflect = 0) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      int dy = FLIP ? 1 - r : r - 1, dx = FLIP ? 1 - sx : sx - 1;
      bool in = (unsigned)(py + dy) < (unsigned)H && (unsigned)(px + dx) < (unsigned)W;
      if (reflect && !in) {
        dy = (py + dy < 0 || py + dy >= H) ? -dy : dy;
        dx = (px + dx < 0 || px + dx >= W) ? -dx : dx;
        in = true;
      }
      ok[r * 3 + sx] = in;
      ptr[r * 3 + sx] = in ? centre + ((long)dy * W + dx) * C : centre;
    }
}

// ---- LayerNorm over C (multiple of 8, <= 512): LPR lanes per row, 64 / LPR rows per wave ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mr,
                                                            long R, int C, int lpr, float eps, const T* __restrict__ add = nullptr,
                                                            const float* __restrict__ scale = nullptr, long rows_per_image = 1, T* __restrict__ xsum = nullptr) {
  // add / scale / xsum (round 6, jg_layernorm_fwd_add): the row that is normalised is x + add * scale[image] -- the residual sum with its
  // DropPath scale, `identity + drop_path(branch)` of a pre-norm transformer block (segformer/backbone.py:401-438) -- written to xsum as well:
  // the sum kernel in front of every LayerNorm (jg_scale with a residual) folded into this pass
  const int lane = threadIdx.x & 63;
  const int sub = lane % lpr, rsub = lane / lpr, rpw = 64 / lpr;
  const long wave = blockIdx.x * 4L + (threadIdx.x >> 6), nwaves = gridDim.x * 4L;
  const bool act = sub * 8 < C;
  float g[8], bt[8];
  if (act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      g[j] = gamma[sub * 8 + j];
      bt[j] = beta[sub * 8 + j];
    }
  }
  for (long r0 = wave * rpw; r0 < R; r0 += nwaves * rpw) {
    const long row = r0 + rsub;
    const bool ok = act && row < R;
    float f[8];
    float s = 0.f, ss = 0.f;
    if (ok) {
      unpack8<T>(*reinterpret_cast<const uint4*>(x + row * C + sub * 8), f);
      if (add) {
        float a8[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(add + row * C + sub * 8), a8);
        const float sc = scale ? scale[row / rows_per_image] : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] += a8[j] * sc;
        const uint4 packed = pack8<T>(f);
        *reinterpret_cast<uint4*>(xsum + row * C + sub * 8) = packed;
        unpack8<T>(packed, f);          // normalise the ROUNDED sum: what a separate sum kernel would have stored and this pass read
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[j];
    }
    for (int o = 1; o < lpr; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    if (ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = f[j] - mean;
        ss += d * d;
      }
    }
    for (int o = 1; o < lpr; o <<= 1) ss += __shfl_xor(ss, o);
    const float rstd = rsqrtf(ss / (float)C + eps);
    if (ok) {
      float o8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] = (f[j] - mean) * rstd * g[j] + bt[j];
      *reinterpret_cast<uint4*>(y + row * C + sub * 8) = pack8<T>(o8);
      if (sub == 0 && mr) {
        mr[row * 2] = mean;
        mr[row * 2 + 1] = rstd;
      }
    }
  }
}

// dx = (res +) rstd (g - mean(g) - xh mean(g xh)),  g = dy gamma,  xh = (x - mean) rstd;  dgamma += sum dy xh;  dbeta += sum dy
// res (optional): the gradient that reaches x through the residual connection around the normalised branch (pre-norm block:
// x + f(LN(x))), added here instead of in a separate pass
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* __restrict__ mr, const T* __restrict__ res, T* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, long R, int C, int lpr,
                                                            const float* __restrict__ scale = nullptr, long rows_per_image = 1, T* __restrict__ dx2 = nullptr) {
  // scale / dx2 (round 6, jg_layernorm_bwd_add2): dx2 = dx * scale[image] as a second output -- the gradient of the DropPath-scaled branch of
  // the residual sum this LayerNorm read (jg_layernorm_fwd_add), which used to be a jg_scale launch over dx
  __shared__ float s_red[2][512];
  const int lane = threadIdx.x & 63;
  const int sub = lane % lpr, rsub = lane / lpr, rpw = 64 / lpr;
  const long wave = blockIdx.x * 4L + (threadIdx.x >> 6), nwaves = gridDim.x * 4L;
  const bool act = sub * 8 < C;
  float g[8], ag[8], ab[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    g[j] = act ? gamma[sub * 8 + j] : 0.f;
    ag[j] = ab[j] = 0.f;
  }
  // two row groups per trip: the eight loads (x, dy, mean, rstd of both) are in flight together -- with one group per trip a wave spent
  // the launch waiting out one HBM round trip after another (8 trips at [131072, 32])
  const float inv_c = 1.0f / (float)C;
  for (long r0 = wave * rpw; r0 < R; r0 += 2 * nwaves * rpw) {
    long row[2];
    bool ok[2];
    uint4 vx[2], vd[2], vr[2];
    float mean[2], rstd[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      row[u] = r0 + u * nwaves * rpw + rsub;
      ok[u] = act && row[u] < R;
      const long rr = ok[u] ? row[u] : 0;            // clamped (valid) address: the loads are unconditional, the results masked
      const int cc = act ? sub * 8 : 0;
      vx[u] = *reinterpret_cast<const uint4*>(x + rr * C + cc);
      vd[u] = *reinterpret_cast<const uint4*>(dy + rr * C + cc);
      vr[u] = res ? *reinterpret_cast<const uint4*>(res + rr * C + cc) : make_uint4(0u, 0u, 0u, 0u);
      mean[u] = mr[rr * 2];
      rstd[u] = mr[rr * 2 + 1];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float xh[8], gy[8], d8[8], f[8];
      float s1 = 0.f, s2 = 0.f;
      unpack8<T>(vx[u], f);
      unpack8<T>(vd[u], d8);
      const float live = ok[u] ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        d8[j] *= live;
        xh[j] = (f[j] - mean[u]) * rstd[u];
        gy[j] = d8[j] * g[j];
        s1 += gy[j];
        s2 += gy[j] * xh[j];
        ag[j] += d8[j] * xh[j];
        ab[j] += d8[j];
      }
      for (int o = 1; o < lpr; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
      }
      if (ok[u] && dx) {
        const float m1 = s1 * inv_c, m2 = s2 * inv_c;
        float o8[8], r8[8];
        unpack8<T>(vr[u], r8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = r8[j] + rstd[u] * (gy[j] - m1 - xh[j] * m2);
        const uint4 packed = pack8<T>(o8);
        *reinterpret_cast<uint4*>(dx + row[u] * C + sub * 8) = packed;
        if (dx2) {          // from the ROUNDED dx, as the separate launch computed it
          const float sc = scale ? scale[row[u] / rows_per_image] : 1.f;
          unpack8<T>(packed, o8);
#pragma unroll
          for (int j = 0; j < 8; ++j) o8[j] *= sc;
          *reinterpret_cast<uint4*>(dx2 + row[u] * C + sub * 8) = pack8<T>(o8);
        }
      }
    }
  }
  if (!dgamma) return;
  // per-channel partials of the block: lanes with the same `sub` (over rows-in-wave and the 4 waves) add up in LDS
  for (int i = threadIdx.x; i < 1024; i += 256) (&s_red[0][0])[i] = 0.f;
  __syncthreads();
  // the rows of a wave that share a channel octet (lane % lpr, lpr a power of two) are summed by xor-shuffles first
#pragma unroll
  for (int j = 0; j < 8; ++j)
    for (int o = lpr; o < 64; o <<= 1) {
      ag[j] += __shfl_xor(ag[j], o);
      ab[j] += __shfl_xor(ab[j], o);
    }
  if (act && lane < lpr) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&s_red[0][sub * 8 + j], ag[j]);
      atomicAdd(&s_red[1][sub * 8 + j], ab[j]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, s_red[0][c]);
    atomicAdd(dbeta + c, s_red[1][c]);
  }
}

// ---- depth-wise 3x3 (pad 1) + bias (+ GELU): a thread owns ONE 8-channel chunk (its 72 weights stay in registers) and strides
// over pixels; a block covers all chunks of 256 / (C/8) pixels at a time, so a wave reads contiguous channel rows
template <typename T, bool FLIP>
__device__ __forceinline__ void dw_apply(const T* __restrict__ x, const float (&wr)[8][9], const float* bs, long p, int px, int py, int H, int W,
                                         int C, int c8, float* acc, int reflect = 0) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = bs ? bs[j] : 0.f;
  const void* ptr[9];
  bool ok[9];
  dw_window<T, FLIP>(x + p * C + c8 * 8, px, py, H, W, C, ptr, ok, reflect);
  uint4 v[9];
  dw_load9(v, ptr);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float f[8];
    unpack8<T>(v[t], f);
    const float m = ok[t] ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += f[j] * m * wr[j][t];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            T* __restrict__ pre, T* __restrict__ y, int B, int H, int W, int C, int gelu, int reflect) {
  const int c8n = C >> 3;
  const int tpb = 256 / c8n > 0 ? 256 / c8n : 1;
  const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
  if (pl >= tpb) return;
  float wr[8][9], bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bs[j] = bias ? bias[c8 * 8 + j] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[j][t] = w[(c8 * 8 + j) * 9 + t];
  }
  const long npix = (long)B * H * W;
  DwWalk wk(blockIdx.x * (unsigned)tpb + pl, gridDim.x * (unsigned)tpb, W, H);
  for (long p = blockIdx.x * (long)tpb + pl; p < npix; p += (long)gridDim.x * tpb, wk.step()) {
    const int px = wk.px, py = wk.py;
    float acc[8];
    dw_apply<T, false>(x, wr, bs, p, px, py, H, W, C, c8, acc, reflect);
    if (pre) *reinterpret_cast<uint4*>(pre + p * C + c8 * 8) = pack8<T>(acc);
    if (gelu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = gelu_fast(acc[j]);
    }
    *reinterpret_cast<uint4*>(y + p * C + c8 * 8) = pack8<T>(acc);
  }
}

// RUN form (round 6): a thread owns a channel chunk and a run of DW_RX consecutive pixels of ONE image row and slides a 3 x 3 window of
// unpacked values along it -- per output pixel 3 loads and 24 conversions (the new column) instead of 9 and 72, the border handling (zero or
// mirrored rows / columns) resolved once per run / column instead of per tap.  (Per-pixel form above: 65 us per launch at [32, 64, 64, 256]
// for 27 us of bytes and 27 us of vector issue.)
constexpr int DW_RX = 8;
template <typename T>
__device__ __forceinline__ void dw_col_load(const T* __restrict__ img, int xs, bool xok, const int (&ys)[3], const bool (&yok)[3], int W, int C, int c8, uint4 (&v)[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const bool ok = xok && yok[r];
    v[r] = ldg16(img + (ok ? ((long)ys[r] * W + xs) * C + c8 * 8 : 0), ok);
  }
}
template <typename T>
__device__ __forceinline__ void dw_col(const T* __restrict__ img, int xs, bool xok, const int (&ys)[3], const bool (&yok)[3], int W, int C, int c8, float (&col)[3][8]) {
  uint4 v[3];
  dw_col_load<T>(img, xs, xok, ys, yok, W, C, c8, v);
#pragma unroll
  for (int r = 0; r < 3; ++r) unpack8<T>(v[r], col[r]);
}
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_fwd_run_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                T* __restrict__ pre, T* __restrict__ y, int B, int H, int W, int C, int gelu, int reflect) {
  const int c8n = C >> 3;
  const int tpb = 256 / c8n > 0 ? 256 / c8n : 1;
  const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
  if (pl >= tpb) return;
  float wr[8][9], bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bs[j] = bias ? bias[c8 * 8 + j] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[j][t] = w[(c8 * 8 + j) * 9 + t];
  }
  const int rpr = (W + DW_RX - 1) / DW_RX;                 // runs per image row
  const long nruns = (long)B * H * rpr;
  for (long run = blockIdx.x * (long)tpb + pl; run < nruns; run += (long)gridDim.x * tpb) {
    const int rx = (int)(run % rpr);
    const long t = run / rpr;
    const int py = (int)(t % H), b = (int)(t / H);
    const int x0 = rx * DW_RX, x1 = min(W, x0 + DW_RX);
    const T* img = x + (long)b * H * W * C;
    int ys[3];
    bool yok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int yy = py + r - 1;
      bool in = (unsigned)yy < (unsigned)H;
      if (reflect && !in) { yy = yy < 0 ? -yy : 2 * H - 2 - yy; in = true; }
      ys[r] = in ? yy : 0;
      yok[r] = in;
    }
    auto col_src = [&](int xx, int& xs, bool& xok) {
      xok = (unsigned)xx < (unsigned)W;
      xs = xx;
      if (reflect && !xok) { xs = xx < 0 ? -xx : 2 * W - 2 - xx; xok = true; }
      if (!xok) xs = 0;
    };
    float c0[3][8], c1[3][8], c2[3][8];
    int xs;
    bool xok;
    col_src(x0 - 1, xs, xok);
    dw_col<T>(img, xs, xok, ys, yok, W, C, c8, c0);
    col_src(x0, xs, xok);
    dw_col<T>(img, xs, xok, ys, yok, W, C, c8, c1);
    // (a one-column lookahead -- the loads of column px + 2 in flight under the multiply-adds of px -- measured 58.5 against 56.6 us: not kept)
    for (int px = x0; px < x1; ++px) {
      col_src(px + 1, xs, xok);
      dw_col<T>(img, xs, xok, ys, yok, W, C, c8, c2);
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = bs[j];
#pragma unroll
        for (int r = 0; r < 3; ++r) a += c0[r][j] * wr[j][r * 3] + c1[r][j] * wr[j][r * 3 + 1] + c2[r][j] * wr[j][r * 3 + 2];
        acc[j] = a;
      }
      const long p = ((long)b * H + py) * W + px;
      if (pre) *reinterpret_cast<uint4*>(pre + p * C + c8 * 8) = pack8<T>(acc);
      if (gelu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = gelu_fast(acc[j]);
      }
      *reinterpret_cast<uint4*>(y + p * C + c8 * 8) = pack8<T>(acc);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          c0[r][j] = c1[r][j];
          c1[r][j] = c2[r][j];
        }
    }
  }
}

// du = dy gelu'(pre) (written to `du`), dbias += sum du, dw[c][tap] += sum_p du[p][c] x[p + tap][c]; each thread keeps ONE channel
// chunk and strides over pixels, block partials through LDS, one atomic per (channel, tap) per block
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_bwd_w_kernel(const T* __restrict__ x, const T* __restrict__ pre, const T* __restrict__ dy,
                                                              T* __restrict__ du, float* __restrict__ dw, float* __restrict__ dbias,
                                                              float* __restrict__ ws, int B, int H, int W, int C, int gelu, int reflect) {
  __shared__ float s_acc[40 * 257];
  const int c8n = C >> 3;
  const int tpb = 256 / c8n > 0 ? 256 / c8n : 1;          // pixel lanes per block (c8n <= 256)
  const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
  const bool act = pl < tpb;
  // accumulators as fp32 PAIRS (channels 2k, 2k+1): the 72 tap FMAs of a chunk are 36 v_pk_fma_f32; out-of-image taps are zeroed on
  // the raw dwords (4 ANDs per tap) instead of 8 multiplies
  dw_f32x2 aw2[4][9], ab2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ab2[k] = (dw_f32x2)(0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) aw2[k][t] = (dw_f32x2)(0.f);
  }
  const long npix = (long)B * H * W;
  if (act) {
    DwWalk wk(blockIdx.x * (unsigned)tpb + pl, gridDim.x * (unsigned)tpb, W, H);
    for (long p = blockIdx.x * (long)tpb + pl; p < npix; p += (long)gridDim.x * tpb, wk.step()) {
      const int px = wk.px, py = wk.py;
      const void* ptr9[9];
      bool ok[9];
      dw_window<T, false>(x + p * C + c8 * 8, px, py, H, W, C, ptr9, ok, reflect);
      const void* ptr[11];
      ptr[0] = dy + p * C + c8 * 8;
      ptr[1] = gelu ? pre + p * C + c8 * 8 : dy + p * C + c8 * 8;
#pragma unroll
      for (int t = 0; t < 9; ++t) ptr[2 + t] = ptr9[t];
      uint4 v[11];
      dw_load11(v, ptr);        // one round trip per pixel: dy, pre and the 3x3 window of x
      const uint32_t dw_[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, pw_[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
      dw_f32x2 d2[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d2[k] = dw_unpack2<T>(dw_[k]);
        if (gelu) {
          const dw_f32x2 q = dw_unpack2<T>(pw_[k]);
          d2[k].x *= gelu_grad_fast(q.x);
          d2[k].y *= gelu_grad_fast(q.y);
        }
        ab2[k] += d2[k];
      }
      if (du) {
        const float d[8] = {d2[0].x, d2[0].y, d2[1].x, d2[1].y, d2[2].x, d2[2].y, d2[3].x, d2[3].y};
        *reinterpret_cast<uint4*>(du + p * C + c8 * 8) = pack8<T>(d);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const uint32_t mk = ok[t] ? 0xffffffffu : 0u;
        const uint32_t xw[4] = {v[2 + t].x & mk, v[2 + t].y & mk, v[2 + t].z & mk, v[2 + t].w & mk};
#pragma unroll
        for (int k = 0; k < 4; ++k) aw2[k][t] += d2[k] * dw_unpack2<T>(xw[k]);
      }
    }
  }
  // Block reduction over the pixel lanes that share a channel chunk.  Accumulator k = tap * 8 + j (k < 72: weight of channel c8 * 8 + j)
  // or 72 + j (bias).  Every thread parks its 80 values in LDS at [k][thread] (conflict-free), then output (k, c8) sums the `tpb` pixel
  // lanes -- in two rounds of 40 values = 40 KB.  (Before: LDS atomics at [(c8 * 8 + j) * 10 + t], a lane stride of 80 floats = 16 lanes
  // per bank on every one of 80 atomics per thread; the per-block epilogue, not the pixel loop, was what the kernel's time went into.)
  float acc[80];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc[72 + 2 * k] = ab2[k].x;
    acc[72 + 2 * k + 1] = ab2[k].y;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      acc[t * 8 + 2 * k] = aw2[k][t].x;
      acc[t * 8 + 2 * k + 1] = aw2[k][t].y;
    }
  }
  const int n = 80 * c8n;
  constexpr int LD = 257;   // row stride of the parking area: odd, so that both walks below (thread-major and k-major) are conflict-free
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r) __syncthreads();
#pragma unroll
    for (int q = 0; q < 40; ++q) s_acc[q * LD + threadIdx.x] = acc[r * 40 + q];
    __syncthreads();
    if (ws) {   // two-phase form: plain coalesced stores (chunk index fastest), dw_partials_sum_kernel adds the blocks up
      for (int o = threadIdx.x; o < 40 * c8n; o += 256) {
        const int q = o / c8n, cc = o - q * c8n;
        float a = 0.f;
        for (int l = 0; l < tpb; ++l) a += s_acc[q * LD + l * c8n + cc];
        ws[(long)blockIdx.x * n + (r * 40 + q) * c8n + cc] = a;
      }
    } else {    // no workspace: atomics, k fastest -- a wave's 64 destinations lie in the 288-byte weight rows of two channel chunks
      for (int o = threadIdx.x; o < 40 * c8n; o += 256) {
        const int cc = o / 40, q = o - cc * 40;
        float a = 0.f;
        for (int l = 0; l < tpb; ++l) a += s_acc[q * LD + l * c8n + cc];
        const int k = r * 40 + q;
        if (k < 72) {
          if (dw) atomicAdd(dw + (cc * 8 + (k & 7)) * 9 + (k >> 3), a);
        } else if (dbias) atomicAdd(dbias + cc * 8 + (k - 72), a);
      }
    }
  }
}

// Second phase of the depth-wise weight gradient: ws is [G][n], n = 80 * c8n = C * 10, column k * c8n + c8 (k as in the first phase).  With atomics
// straight from the G <= 1024 blocks of the first phase every one of the n addresses carries a G-deep same-address chain (measured: the
// kernel took 54 - 135 us for 3 - 27 us of HBM traffic); here a block sums 64 rows of 64 columns and the chains are G / 64 deep.
__global__ __launch_bounds__(256) void dw_partials_sum_kernel(const float* __restrict__ ws, int G, int n, int c8n, float* __restrict__ dw,
                                                              float* __restrict__ dbias) {
  __shared__ float s_p[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + l;
  const int g0 = blockIdx.y * 64 + w * 16;
  float a = 0.f;
  if (i < n) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int g = g0 + k;
      if (g < G) a += ws[(long)g * n + i];
    }
  }
  s_p[w][l] = a;
  __syncthreads();
  if (w == 0 && i < n) {
    a = (s_p[0][l] + s_p[1][l]) + (s_p[2][l] + s_p[3][l]);
    const int k = i / c8n, cc = i - k * c8n;
    if (k < 72) {
      if (dw) atomicAdd(dw + (cc * 8 + (k & 7)) * 9 + (k >> 3), a);
    } else if (dbias) atomicAdd(dbias + cc * 8 + (k - 72), a);
  }
}

// dx[p][c] = sum_taps du[p - tap][c] w[c][tap]   (same thread layout as the forward)
// reflect: x was read through the mirror, so pixel q also collects what the padded-domain gradient holds at its mirror images -- row -1 for
// q in row 1, row H for row H - 2, likewise the columns (up to four windows in the corners): dx[q] = sum_{q' -> q} sum_taps du[q' - tap] w[tap],
// du zero outside the image.  The extra windows are centred OUTSIDE the image; their taps are addressed from the image origin.
template <typename T>
__device__ __forceinline__ void dw_apply_flip_at(const T* __restrict__ img, const float (&wr)[8][9], int yc, int xc, int H, int W, int C, int c8, float* acc) {
  const void* ptr[9];
  bool ok[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      const int yy = yc + 1 - r, xx = xc + 1 - sx;
      const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      ok[r * 3 + sx] = in;
      ptr[r * 3 + sx] = img + (in ? ((long)yy * W + xx) * C : 0) + c8 * 8;
    }
  uint4 v[9];
  dw_load9(v, ptr);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float f[8];
    unpack8<T>(v[t], f);
    const float m = ok[t] ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += f[j] * m * wr[j][t];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_bwd_x_kernel(const T* __restrict__ du, const float* __restrict__ w, T* __restrict__ dx, int B,
                                                              int H, int W, int C, int reflect) {
  const int c8n = C >> 3;
  const int tpb = 256 / c8n > 0 ? 256 / c8n : 1;
  const int c8 = threadIdx.x % c8n, pl = threadIdx.x / c8n;
  if (pl >= tpb) return;
  float wr[8][9];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[j][t] = w[(c8 * 8 + j) * 9 + t];
  const long npix = (long)B * H * W;
  DwWalk wk(blockIdx.x * (unsigned)tpb + pl, gridDim.x * (unsigned)tpb, W, H);
  for (long p = blockIdx.x * (long)tpb + pl; p < npix; p += (long)gridDim.x * tpb, wk.step()) {
    const int px = wk.px, py = wk.py;
    float acc[8];
    dw_apply<T, true>(du, wr, nullptr, p, px, py, H, W, C, c8, acc);
    if (reflect) {
      const int ym = py == 1 ? -1 : -2, yM = py == H - 2 ? H : -2;      // mirror rows of this pixel's row (-2: none)
      const int xm = px == 1 ? -1 : -2, xM = px == W - 2 ? W : -2;
      if (ym != -2 || yM != -2 || xm != -2 || xM != -2) {              // the ring next to the border only
        const T* img = du + (p - ((long)py * W + px)) * C;
        const int ys[3] = {py, ym, yM}, xs[3] = {px, xm, xM};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2)
            if ((a | b2) && ys[a] != -2 && xs[b2] != -2) dw_apply_flip_at<T>(img, wr, ys[a], xs[b2], H, W, C, c8, acc);
      }
    }
    *reinterpret_cast<uint4*>(dx + p * C + c8 * 8) = pack8<T>(acc);
  }
}

// ---- attention with a small key/value set (head dim 32): block = (64 queries, head, image), K/V of the head in LDS ----------
constexpr int AKV_MAX = 256;
// a . row and acc += w * row for a 32-float LDS row that every lane reads at the same address (8 broadcast ds_read_b128)
__device__ __forceinline__ float dot32(const float* a, const float* row) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four independent chains: one wave per SIMD has nothing else to hide FMA latency
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 r = *reinterpret_cast<const float4*>(row + 4 * c);
    s0 += a[4 * c] * r.x;
    s1 += a[4 * c + 1] * r.y;
    s2 += a[4 * c + 2] * r.z;
    s3 += a[4 * c + 3] * r.w;
  }
  return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ void axpy32(float* acc, float w, const float* row) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 r = *reinterpret_cast<const float4*>(row + 4 * c);
    acc[4 * c] += w * r.x;
    acc[4 * c + 1] += w * r.y;
    acc[4 * c + 2] += w * r.z;
    acc[4 * c + 3] += w * r.w;
  }
}
template <typename T, int KVMAX>
__global__ __launch_bounds__(64) void attn_smallkv_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                              T* __restrict__ o, float* __restrict__ lse, int Tq, int Tkv, int heads, long ldq,
                                                              long ldkv, long ldo, float scale) {
  __shared__ __attribute__((aligned(16))) float s_k[KVMAX][32], s_v[KVMAX][32];   // rows are read as broadcast float4s
  const int h = blockIdx.y, b = blockIdx.z, t = threadIdx.x;
  for (int i = t; i < Tkv * 4; i += 64) {
    const int j = i >> 2, c = (i & 3) * 8;
    float f[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(k + ((long)b * Tkv + j) * ldkv + h * 32 + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) s_k[j][c + e] = f[e];
    unpack8<T>(*reinterpret_cast<const uint4*>(v + ((long)b * Tkv + j) * ldkv + h * 32 + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) s_v[j][c + e] = f[e];
  }
  __syncthreads();
  const int qi = blockIdx.x * 64 + t;
  if (qi >= Tq) return;
  float qv[32];
#pragma unroll
  for (int c = 0; c < 4; ++c) unpack8<T>(*reinterpret_cast<const uint4*>(q + ((long)b * Tq + qi) * ldq + h * 32 + c * 8), qv + c * 8);
  float m = -3.0e38f;
  for (int j = 0; j < Tkv; ++j) m = fmaxf(m, dot32(qv, s_k[j]) * scale);
  float l = 0.f, acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
  for (int j = 0; j < Tkv; ++j) {
    const float p = expf(dot32(qv, s_k[j]) * scale - m);
    l += p;
    axpy32(acc, p, s_v[j]);
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] *= inv;
#pragma unroll
  for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4*>(o + ((long)b * Tq + qi) * ldo + h * 32 + c * 8) = pack8<T>(acc + c * 8);
  if (lse) lse[((long)b * heads + h) * Tq + qi] = m + logf(l);
}

// dq per query thread.  dk / dv: every 32-query half of the block parks p and ds of one 64-key slab in LDS ([key][query]), then
// the threads switch roles -- thread = key -- and contract them against the block's q / dO rows (read back from global memory: all
// lanes want the same row, a broadcast); no LDS atomics; one fp32 global atomic per (key, channel) and block into dkf / dvf.
// LDS = K, V + two [64][33] slabs: 34 KB at T_kv <= 64, i.e. 4 workgroups per CU instead of 2.
template <typename T, int KVMAX>
__global__ __launch_bounds__(64) void attn_smallkv_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                              const T* __restrict__ o, const T* __restrict__ dout, const float* __restrict__ lse,
                                                              T* __restrict__ dq, float* __restrict__ dkf, float* __restrict__ dvf, int Tq, int Tkv,
                                                              int heads, long ldq, long ldkv, long ldo, float scale) {
  __shared__ __attribute__((aligned(16))) float s_k[KVMAX][32], s_v[KVMAX][32];   // rows are read as broadcast float4s
  __shared__ float s_p[64][33], s_ds[64][33];
  const int h = blockIdx.y, b = blockIdx.z, t = threadIdx.x;
  for (int i = t; i < Tkv * 4; i += 64) {
    const int j = i >> 2, c = (i & 3) * 8;
    float f[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(k + ((long)b * Tkv + j) * ldkv + h * 32 + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) s_k[j][c + e] = f[e];
    unpack8<T>(*reinterpret_cast<const uint4*>(v + ((long)b * Tkv + j) * ldkv + h * 32 + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) s_v[j][c + e] = f[e];
  }
  // T_kv <= 64 (one key slab): a block walks QT query tiles and flushes dk / dv once -- the fp32 atomics onto the small
  // [B][T_kv][C] arrays were the bottleneck (every address is hit by T_q / 64 blocks)
  constexpr int QT = 1;      // (4 tiles per block were measured slower: fewer, longer workgroups; kept for T_q >> 4096)
  const int H32 = heads * 32;
  float dk[32], dv[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) dk[c] = dv[c] = 0.f;
  for (int tile = 0; tile < QT; ++tile) {
    const int q0 = (blockIdx.x * QT + tile) * 64;
    if (q0 >= Tq) break;
    const int qi = q0 + t;
    const bool ok = qi < Tq;
    float qv[32], dov[32], dqv[32];
    float D = 0.f, L = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) qv[c] = dov[c] = dqv[c] = 0.f;
    if (ok) {
      float ov[32];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unpack8<T>(*reinterpret_cast<const uint4*>(q + ((long)b * Tq + qi) * ldq + h * 32 + c * 8), qv + c * 8);
        unpack8<T>(*reinterpret_cast<const uint4*>(dout + ((long)b * Tq + qi) * ldo + h * 32 + c * 8), dov + c * 8);
        unpack8<T>(*reinterpret_cast<const uint4*>(o + ((long)b * Tq + qi) * ldo + h * 32 + c * 8), ov + c * 8);
      }
#pragma unroll
      for (int c = 0; c < 32; ++c) D += dov[c] * ov[c];
      L = lse[((long)b * heads + h) * Tq + qi];
    }
    for (int j0 = 0; j0 < Tkv; j0 += 64) {
      const int jn = min(64, Tkv - j0);
      if (QT == 1) {
#pragma unroll
        for (int c = 0; c < 32; ++c) dk[c] = dv[c] = 0.f;
      }
      for (int half = 0; half < 2; ++half) {
        __syncthreads();                    // K / V visible; the previous role-switched reads of s_p / s_ds are done
        for (int jj = 0; jj < jn; ++jj) {
          const int j = j0 + jj;
          const float s = dot32(qv, s_k[j]), dp = dot32(dov, s_v[j]);
          const float p = ok ? expf(s * scale - L) : 0.f;
          const float ds = p * (dp - D) * scale;
          if (half == 0) axpy32(dqv, ds, s_k[j]);
          if ((t >> 5) == half) {
            s_p[jj][t & 31] = p;
            s_ds[jj][t & 31] = ds;
          }
        }
        __syncthreads();
        if (t < jn) {                       // thread = key j0 + t, over the 32 queries of this half
          for (int u = 0; u < 32; ++u) {
            const int qu = q0 + half * 32 + u;
            if (qu >= Tq) break;
            const float pv = s_p[t][u], dsv = s_ds[t][u];
            float qr[32], dr[32];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              unpack8<T>(*reinterpret_cast<const uint4*>(q + ((long)b * Tq + qu) * ldq + h * 32 + c * 8), qr + c * 8);
              unpack8<T>(*reinterpret_cast<const uint4*>(dout + ((long)b * Tq + qu) * ldo + h * 32 + c * 8), dr + c * 8);
            }
#pragma unroll
            for (int c = 0; c < 32; ++c) {
              dk[c] += dsv * qr[c];
              dv[c] += pv * dr[c];
            }
          }
        }
      }
      if (QT == 1 && t < jn) {
        float* dkp = dkf + ((long)b * Tkv + j0 + t) * H32 + h * 32;
        float* dvp = dvf + ((long)b * Tkv + j0 + t) * H32 + h * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
          atomicAdd(dkp + c, dk[c]);
          atomicAdd(dvp + c, dv[c]);
        }
      }
    }
    if (ok) {
#pragma unroll
      for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4*>(dq + ((long)b * Tq + qi) * ldq + h * 32 + c * 8) = pack8<T>(dqv + c * 8);
    }
  }
  if (QT > 1 && t < Tkv) {
    float* dkp = dkf + ((long)b * Tkv + t) * H32 + h * 32;
    float* dvp = dvf + ((long)b * Tkv + t) * H32 + h * 32;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      atomicAdd(dkp + c, dk[c]);
      atomicAdd(dvp + c, dv[c]);
    }
  }
}

// ---- bilinear resize, align_corners = False (aten upsample_bilinear2d): src = (dst + 0.5) * in / out - 0.5, clamped at 0 ------------
// ALIGN (align_corners = True, FeatureFusionBlockMatrix of the projected discriminator): src = dst * (in - 1) / (out - 1)
template <bool ALIGN = false>
__device__ __forceinline__ void bil_coord(int o, int in, int out, int& i0, int& i1, float& w1) {
  float s = ALIGN ? (out > 1 ? (float)o * ((float)(in - 1) / (float)(out - 1)) : 0.f) : ((float)o + 0.5f) * ((float)in / (float)out) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = s - (float)i0;
}
template <typename T, bool ALIGN = false>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo, long ldy) {
  const int c8n = C >> 3;
  const long total = (long)B * Ho * Wo * c8n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = i % c8n;
    long p = i / c8n;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho, b = p / Ho;
    int y0, y1, x0, x1;
    float wy, wx;
    bil_coord<ALIGN>(oy, H, Ho, y0, y1, wy);
    bil_coord<ALIGN>(ox, W, Wo, x0, x1, wx);
    float a[8], bq[8], c[8], d[8], o8[8];
    const T* xb = x + (long)b * H * W * C + c8 * 8;
    unpack8<T>(*reinterpret_cast<const uint4*>(xb + ((long)y0 * W + x0) * C), a);
    unpack8<T>(*reinterpret_cast<const uint4*>(xb + ((long)y0 * W + x1) * C), bq);
    unpack8<T>(*reinterpret_cast<const uint4*>(xb + ((long)y1 * W + x0) * C), c);
    unpack8<T>(*reinterpret_cast<const uint4*>(xb + ((long)y1 * W + x1) * C), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = (1.f - wy) * ((1.f - wx) * a[j] + wx * bq[j]) + wy * ((1.f - wx) * c[j] + wx * d[j]);
    *reinterpret_cast<uint4*>(y + (((long)b * Ho + oy) * Wo + ox) * ldy + c8 * 8) = pack8<T>(o8);
  }
}
// y = act(x0 + sum_i bilinear(x_i -> Ho x Wo)), up to three resized terms (round 6): the SegFormer head with its fusion convolution applied
// BEFORE the resize (SegformerHead.forward, modules/segformer.py) sums four 256-channel maps instead of concatenating them
struct ResizeSumP {
  const void* x[3];
  int H[3], W[3];
  int n;
};
template <typename T>
__global__ void resize_sum_kernel(const T* __restrict__ x0, ResizeSumP rp, T* __restrict__ y, int B, int C, int Ho, int Wo, int act,
                                  const float* __restrict__ chs) {
  const int c8n = C >> 3;
  const long total = (long)B * Ho * Wo * c8n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = i % c8n;
    long p = i / c8n;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho, b = p / Ho;
    float o8[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(x0 + i * 8), o8);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t >= rp.n) break;
      const int H = rp.H[t], W = rp.W[t];
      int y0, y1, x0i, x1i;
      float wy, wx;
      bil_coord<false>(oy, H, Ho, y0, y1, wy);
      bil_coord<false>(ox, W, Wo, x0i, x1i, wx);
      float a[8], bq[8], c[8], d[8];
      const T* xb = (const T*)rp.x[t] + (long)b * H * W * C + c8 * 8;
      unpack8<T>(*reinterpret_cast<const uint4*>(xb + ((long)y0 * W + x0i) * C), a);
      unpack8<T>(*reinterpret_cast<const uint4*>(xb + ((long)y0 * W + x1i) * C), bq);
      unpack8<T>(*reinterpret_cast<const uint4*>(xb + ((long)y1 * W + x0i) * C), c);
      unpack8<T>(*reinterpret_cast<const uint4*>(xb + ((long)y1 * W + x1i) * C), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o8[j] += (1.f - wy) * ((1.f - wx) * a[j] + wx * bq[j]) + wy * ((1.f - wx) * c[j] + wx * d[j]);
      }
    }
    if (act == JG_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] = fmaxf(o8[j], 0.f);
    }
    if (chs) {      // nn.Dropout2d behind the head's ReLU as a per-(image, channel) factor >= 0 (round 6: no pass of its own)
#pragma unroll
      for (int j = 0; j < 8; ++j) o8[j] *= chs[(long)b * C + c8 * 8 + j];
    }
    *reinterpret_cast<uint4*>(y + i * 8) = pack8<T>(o8);
  }
}
// adjoint as a gather: an input pixel collects from the output pixels whose two source rows / columns include it
template <typename T, bool ALIGN = false>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo, long lddy) {
  const int c8n = C >> 3;
  const long total = (long)B * H * W * c8n;
  // output rows whose source coordinate lies in (iy - 1, iy + 1): from the REAL scale (any ratio >= 1, e.g. 64 -> 96; round 5: the window
  // used to be derived from ceil(out / in), which is only right for integer ratios), one output pixel of margin each way
  const float ry = ALIGN ? (H > 1 ? (float)(Ho - 1) / (float)(H - 1) : (float)Ho) : (float)Ho / (float)H;
  const float rx = ALIGN ? (W > 1 ? (float)(Wo - 1) / (float)(W - 1) : (float)Wo) : (float)Wo / (float)W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = i % c8n;
    long p = i / c8n;
    const int ix = p % W;
    p /= W;
    const int iy = p % H, b = p / H;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const float offl = ALIGN ? -1.0f : -0.5f, offh = ALIGN ? 1.0f : 1.5f, sh = ALIGN ? 0.f : 0.5f;
    const int oy_lo = max(0, (int)floorf(((float)iy + offl) * ry - sh) - 1), oy_hi = min(Ho - 1, (int)ceilf(((float)iy + offh) * ry - sh) + 1);
    const int ox_lo = max(0, (int)floorf(((float)ix + offl) * rx - sh) - 1), ox_hi = min(Wo - 1, (int)ceilf(((float)ix + offh) * rx - sh) + 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float wy;
      bil_coord<ALIGN>(oy, H, Ho, y0, y1, wy);
      const float ky = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
      if (ky == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float wx;
        bil_coord<ALIGN>(ox, W, Wo, x0, x1, wx);
        const float kx = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
        if (kx == 0.f) continue;
        float f[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(dy + (((long)b * Ho + oy) * Wo + ox) * lddy + c8 * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += ky * kx * f[j];
      }
    }
    *reinterpret_cast<uint4*>(dx + (((long)b * H + iy) * W + ix) * C + c8 * 8) = pack8<T>(acc);
  }
}

// ---- adjoint of resize_sum_kernel in separable form (round 6, jg_resize_sum_bwd) --------------------------------------------------------
// The gather kernel above walks a (2r + 3)^2 window of dy per input pixel: at ratio 8 (the 8 x 8 map of the SegFormer head against 64 x 64)
// 361 dependent trips on 65 K threads -- 112 us for 67 MB, and three such launches behind an activation-gradient pass read dy four times.
// Here dx_s = R_y^T (R_x^T g), g = dy act'(y): pass X (one workgroup per output row: the row of g in LDS, written out once as the gradient
// of the full-resolution term) reduces along x for ALL terms, pass Y reduces the fp32 intermediates along y.  Two launches, dy read once.
struct ResizeBwdP {
  void* dx[3];
  int H[3], W[3];
  long off[3];      // float offset of a term's intermediate [B, Ho, W_s, C] in the workspace
  int n;
};
__device__ __forceinline__ void bil_window(int i, int in, int out, int& lo, int& hi) {      // output indices whose two taps may include input index i
  const float r = (float)out / (float)in;
  lo = max(0, (int)floorf(((float)i - 0.5f) * r - 0.5f) - 1);
  hi = min(out - 1, (int)ceilf(((float)i + 1.5f) * r - 0.5f) + 1);
}
template <typename T>
__global__ __launch_bounds__(256) void resize_sum_bwd_x_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ g, ResizeBwdP rp,
                                                               float* __restrict__ ws, int C, int Ho, int Wo, int act, const float* __restrict__ chs) {
  extern __shared__ uint4 s_row[];      // [Wo][C / 8] 16-byte chunks of g
  const int c8n = C >> 3, tid = threadIdx.x;
  const long row = blockIdx.x;          // b * Ho + oy
  const int nrow = Wo * c8n;
  for (int i = tid; i < nrow; i += 256) {
    uint4 v = *reinterpret_cast<const uint4*>(dy + (row * Wo) * C + (long)i * 8);
    if (act == JG_ACT_RELU || chs) {
      float f[8], d[8];
      unpack8<T>(v, d);
      if (chs) {      // y = relu(.) * s with s >= 0: y > 0 exactly where the ReLU passed AND the channel was kept
        const float* sc = chs + (row / Ho) * C + (i % c8n) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] *= sc[q];
      }
      if (act == JG_ACT_RELU) {
        unpack8<T>(*reinterpret_cast<const uint4*>(y + (row * Wo) * C + (long)i * 8), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = f[q] > 0.f ? d[q] : 0.f;
      }
      v = pack8<T>(d);
    }
    if (g) *reinterpret_cast<uint4*>(g + (row * Wo) * C + (long)i * 8) = v;
    s_row[i] = v;
  }
  __syncthreads();
  for (int t = 0; t < rp.n; ++t) {
    const int W = rp.W[t], items = W * c8n;
    float* wt = ws + rp.off[t] + row * (long)W * C;
    for (int it = tid; it < items; it += 256) {
      const int ix = it / c8n, c8 = it - ix * c8n;
      int lo, hi;
      bil_window(ix, W, Wo, lo, hi);
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      for (int ox = lo; ox <= hi; ++ox) {
        int x0, x1;
        float wx;
        bil_coord<false>(ox, W, Wo, x0, x1, wx);
        const float kx = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
        if (kx == 0.f) continue;
        float f[8];
        unpack8<T>(s_row[ox * c8n + c8], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += kx * f[j];
      }
      float4* o = reinterpret_cast<float4*>(wt + (long)ix * C + c8 * 8);
      o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void resize_sum_bwd_y_kernel(const float* __restrict__ ws, ResizeBwdP rp, int B, int C, int Ho) {
  const int c8n = C >> 3;
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  int t = 0;
  for (; t < rp.n; ++t) {
    const long cnt = (long)B * rp.H[t] * rp.W[t] * c8n;
    if (i < cnt) break;
    i -= cnt;
  }
  if (t >= rp.n) return;
  const int H = rp.H[t], W = rp.W[t];
  const int c8 = i % c8n;
  long p = i / c8n;
  const int ix = p % W;
  p /= W;
  const int iy = p % H, b = p / H;
  int lo, hi;
  bil_window(iy, H, Ho, lo, hi);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const float* wt = ws + rp.off[t] + ((long)b * Ho * W + ix) * C + c8 * 8;
  for (int oy = lo; oy <= hi; ++oy) {
    int y0, y1;
    float wy;
    bil_coord<false>(oy, H, Ho, y0, y1, wy);
    const float ky = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
    if (ky == 0.f) continue;
    const float4 a = *reinterpret_cast<const float4*>(wt + (long)oy * W * C), c = *reinterpret_cast<const float4*>(wt + (long)oy * W * C + 4);
    acc[0] += ky * a.x; acc[1] += ky * a.y; acc[2] += ky * a.z; acc[3] += ky * a.w;
    acc[4] += ky * c.x; acc[5] += ky * c.y; acc[6] += ky * c.z; acc[7] += ky * c.w;
  }
  *reinterpret_cast<uint4*>((T*)rp.dx[t] + (((long)b * H + iy) * W + ix) * C + c8 * 8) = pack8<T>(acc);
}

// ---- BatchNorm2d coefficients from the per-(image, channel) sums of jg_gn_stats -----------------------------------------------
__global__ void bn_coef_kernel(const float* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                               float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ ab, float* __restrict__ mr,
                               int B, int HW, int C, float eps, float momentum, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    float s = 0.f, ss = 0.f;
    for (int b = 0; b < B; ++b) {
      s += sums[((long)b * C + c) * 2];
      ss += sums[((long)b * C + c) * 2 + 1];
    }
    const float n = (float)B * (float)HW;
    mean = s / n;
    var = fmaxf(ss / n - mean * mean, 0.f);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * n / (n - 1.f);
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float rstd = rsqrtf(var + eps);
  const float a = gamma[c] * rstd, bb = beta[c] - mean * a;
  for (int b = 0; b < B; ++b) {
    ab[((long)b * C + c) * 2] = a;
    ab[((long)b * C + c) * 2 + 1] = bb;
  }
  mr[c * 2] = mean;
  mr[c * 2 + 1] = rstd;
}
// red[b][c] = (sum du, sum du x) -> dx = du P + x Q + R with batch statistics; dgamma += sum du xh; dbeta += sum du
__global__ void bn_bwd_coef_kernel(const float* __restrict__ red, const float* __restrict__ gamma, const float* __restrict__ mr,
                                   float* __restrict__ pqr, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int HW, int C,
                                   int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    s1 += red[((long)b * C + c) * 2];
    s2 += red[((long)b * C + c) * 2 + 1];
  }
  const float mean = mr[c * 2], rstd = mr[c * 2 + 1], g = gamma[c];
  const float sxh = (s2 - mean * s1) * rstd;          // sum du xh
  if (dgamma) atomicAdd(dgamma + c, sxh);
  if (dbeta) atomicAdd(dbeta + c, s1);
  float P = g * rstd, Q = 0.f, R = 0.f;
  if (training) {
    const float n = (float)B * (float)HW;
    // dx = g rstd (du - s1/n - xh sxh/n),  xh = (x - mean) rstd
    Q = -g * rstd * rstd * sxh / n;
    R = -g * rstd * s1 / n - Q * mean;
  }
  for (int b = 0; b < B; ++b) {
    pqr[((long)b * C + c) * 3] = P;
    pqr[((long)b * C + c) * 3 + 1] = Q;
    pqr[((long)b * C + c) * 3 + 2] = R;
  }
}

// ---- attention composition (attn_network.py:14-46, segformer_generator.py:141-164) ------------------------------------------------
// thread per attention cell (f x f full-resolution pixels): a = softmax(logits[0 .. na)), out[c] = sum_{i < ni} img[nc i + c] a_i +
// sum_{i >= ni} xin[c] a_i
constexpr int ACOMP_MAX = 16;
template <typename T>
__global__ void attn_compose_fwd_kernel(const T* __restrict__ img, const T* __restrict__ logits, const T* __restrict__ xin, T* __restrict__ out,
                                        int B, int S, int f, int na, int ni, int nc, int ldimg, int ldl, int ldx, int ldo) {
  const int Sa = S / f;
  const long total = (long)B * Sa * Sa;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ax = i % Sa, ay = (i / Sa) % Sa, b = i / ((long)Sa * Sa);
    float a[ACOMP_MAX];
    float m = -3.0e38f, l = 0.f;
    for (int k = 0; k < na; ++k) {
      a[k] = to_f32(logits[i * ldl + k]);
      m = fmaxf(m, a[k]);
    }
    for (int k = 0; k < na; ++k) {
      a[k] = expf(a[k] - m);
      l += a[k];
    }
    float ain = 0.f;
    for (int k = 0; k < na; ++k) {
      a[k] /= l;
      if (k >= ni) ain += a[k];
    }
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) {
        const long p = ((long)b * S + ay * f + dy) * S + ax * f + dx;
        for (int c = 0; c < ldo; ++c) {
          float v = 0.f;
          if (c < nc) {
            v = to_f32(xin[p * ldx + c]) * ain;
            for (int k = 0; k < ni; ++k) v += to_f32(img[p * ldimg + nc * k + c]) * a[k];
          }
          out[p * ldo + c] = from_f32<T>(v);
        }
      }
  }
}
template <typename T>
__global__ void attn_compose_bwd_kernel(const T* __restrict__ img, const T* __restrict__ logits, const T* __restrict__ xin, const T* __restrict__ dout,
                                        T* __restrict__ dimg, T* __restrict__ dlogits, T* __restrict__ dxin, int B, int S, int f, int na, int ni,
                                        int nc, int ldimg, int ldl, int ldx, int ldo) {
  const int Sa = S / f;
  const long total = (long)B * Sa * Sa;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ax = i % Sa, ay = (i / Sa) % Sa, b = i / ((long)Sa * Sa);
    float a[ACOMP_MAX], da[ACOMP_MAX];
    float m = -3.0e38f, l = 0.f;
    for (int k = 0; k < na; ++k) {
      a[k] = to_f32(logits[i * ldl + k]);
      m = fmaxf(m, a[k]);
      da[k] = 0.f;
    }
    for (int k = 0; k < na; ++k) {
      a[k] = expf(a[k] - m);
      l += a[k];
    }
    float ain = 0.f;
    for (int k = 0; k < na; ++k) {
      a[k] /= l;
      if (k >= ni) ain += a[k];
    }
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) {
        const long p = ((long)b * S + ay * f + dy) * S + ax * f + dx;
        float dain = 0.f;
        for (int c = 0; c < nc; ++c) {
          const float g = to_f32(dout[p * ldo + c]);
          for (int k = 0; k < ni; ++k) {
            da[k] += g * to_f32(img[p * ldimg + nc * k + c]);
            dimg[p * ldimg + nc * k + c] = from_f32<T>(g * a[k]);
          }
          dain += g * to_f32(xin[p * ldx + c]);
          if (dxin) dxin[p * ldx + c] = from_f32<T>(g * ain);
        }
        for (int k = ni; k < na; ++k) da[k] += dain;
        for (int c = nc * ni; c < ldimg; ++c) dimg[p * ldimg + c] = from_f32<T>(0.f);
        if (dxin)
          for (int c = nc; c < ldx; ++c) dxin[p * ldx + c] = from_f32<T>(0.f);
      }
    float dot = 0.f;
    for (int k = 0; k < na; ++k) dot += a[k] * da[k];
    for (int k = 0; k < ldl; ++k) dlogits[i * ldl + k] = from_f32<T>(k < na ? a[k] * (da[k] - dot) : 0.f);
  }
}

// pixel-parallel forms of the two kernels above (one lane per PIXEL, the f x f lanes of an attention cell side by side; 16-byte loads /
// stores of the pixel's channels; the cell's logit gradient is a shuffle reduction over its lanes).  The cell-per-thread kernels walk
// f^2 pixels x 35 scattered 2-byte accesses per thread: 1.0 ms / 0.3 ms per launch at 32 x 256^2 against ~0.1 ms of HBM time.
// Requirements: f in {1, 2, 4, 8}, ldimg / ldx / ldo / ldl multiples of 8, ldimg <= 64, ldl <= 16, nc <= 8.
template <typename T>
__device__ __forceinline__ void acomp_softmax(const T* __restrict__ lrow, int ldl, int na, int ni, float* a, float& ain) {
  float lg[16];
  unpack8<T>(*reinterpret_cast<const uint4*>(lrow), lg);
  if (ldl > 8) unpack8<T>(*reinterpret_cast<const uint4*>(lrow + 8), lg + 8);
  float m = -3.0e38f, l = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k < na) m = fmaxf(m, lg[k]);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    a[k] = k < na ? expf(lg[k] - m) : 0.f;
    l += a[k];
  }
  ain = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    a[k] /= l;
    if (k >= ni && k < na) ain += a[k];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_compose_fwd_px_kernel(const T* __restrict__ img, const T* __restrict__ logits, const T* __restrict__ xin,
                                                                   T* __restrict__ out, int B, int S, int f, int na, int ni, int nc, int ldimg,
                                                                   int ldl, int ldx, int ldo) {
  const int Sa = S / f, ff = f * f;
  const long total = (long)B * S * S;
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long cell = t / ff;
  const int sub = (int)(t % ff), dy = sub / f, dx = sub % f;
  const int ax = cell % Sa, ay = (cell / Sa) % Sa, b = (int)(cell / ((long)Sa * Sa));
  const long p = ((long)b * S + ay * f + dy) * S + ax * f + dx;
  float a[16], ain;
  acomp_softmax<T>(logits + cell * ldl, ldl, na, ni, a, ain);
  float xv[8], o[8];
  unpack8<T>(*reinterpret_cast<const uint4*>(xin + p * ldx), xv);
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = c < nc ? xv[c] * ain : 0.f;
  for (int j = 0; j < ldimg / 8; ++j) {
    float v[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(img + p * ldimg + j * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = j * 8 + e;
      if (ci < nc * ni) {
        const int k = ci / nc, c = ci - k * nc;
        float ak = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) ak = q == k ? a[q] : ak;      // register array: select, do not index
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] += q == c ? v[e] * ak : 0.f;
      }
    }
  }
  *reinterpret_cast<uint4*>(out + p * ldo) = pack8<T>(o);
  for (int j = 1; j < ldo / 8; ++j) *reinterpret_cast<uint4*>(out + p * ldo + j * 8) = make_uint4(0u, 0u, 0u, 0u);
}

template <typename T>
__global__ __launch_bounds__(256) void attn_compose_bwd_px_kernel(const T* __restrict__ img, const T* __restrict__ logits, const T* __restrict__ xin,
                                                                   const T* __restrict__ dout, T* __restrict__ dimg, T* __restrict__ dlogits,
                                                                   T* __restrict__ dxin, int B, int S, int f, int na, int ni, int nc, int ldimg,
                                                                   int ldl, int ldx, int ldo) {
  const int Sa = S / f, ff = f * f;
  const long total = (long)B * S * S;       // a multiple of ff, and ff divides 64: a cell never straddles waves or the grid end
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long cell = t / ff;
  const int sub = (int)(t % ff), dy = sub / f, dx = sub % f;
  const int ax = cell % Sa, ay = (cell / Sa) % Sa, b = (int)(cell / ((long)Sa * Sa));
  const long p = ((long)b * S + ay * f + dy) * S + ax * f + dx;
  float a[16], ain;
  acomp_softmax<T>(logits + cell * ldl, ldl, na, ni, a, ain);
  float g[8], xv[8], da[16];
  unpack8<T>(*reinterpret_cast<const uint4*>(dout + p * ldo), g);
  unpack8<T>(*reinterpret_cast<const uint4*>(xin + p * ldx), xv);
#pragma unroll
  for (int k = 0; k < 16; ++k) da[k] = 0.f;
  float dain = 0.f, dx8[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c >= nc) g[c] = 0.f;
    dain += g[c] * xv[c];
    dx8[c] = g[c] * ain;
  }
  if (dxin) {
    *reinterpret_cast<uint4*>(dxin + p * ldx) = pack8<T>(dx8);
    for (int j = 1; j < ldx / 8; ++j) *reinterpret_cast<uint4*>(dxin + p * ldx + j * 8) = make_uint4(0u, 0u, 0u, 0u);
  }
  for (int j = 0; j < ldimg / 8; ++j) {
    float v[8], d8[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(img + p * ldimg + j * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = j * 8 + e;
      d8[e] = 0.f;
      if (ci < nc * ni) {
        const int k = ci / nc, c = ci - k * nc;
        float gc = 0.f, ak = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) gc = q == c ? g[q] : gc;
#pragma unroll
        for (int q = 0; q < 16; ++q) ak = q == k ? a[q] : ak;
        d8[e] = gc * ak;
#pragma unroll
        for (int q = 0; q < 16; ++q) da[q] += q == k ? gc * v[e] : 0.f;
      }
    }
    *reinterpret_cast<uint4*>(dimg + p * ldimg + j * 8) = pack8<T>(d8);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k >= ni && k < na) da[k] += dain;
    for (int o = 1; o < ff; o <<= 1) da[k] += __shfl_xor(da[k], o);      // the cell's lanes are an aligned group of ff
  }
  if (sub == 0) {
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) dot += a[k] * da[k];
    float dl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) dl[k] = k < na ? a[k] * (da[k] - dot) : 0.f;
    *reinterpret_cast<uint4*>(dlogits + cell * ldl) = pack8<T>(dl);
    if (ldl > 8) *reinterpret_cast<uint4*>(dlogits + cell * ldl + 8) = pack8<T>(dl + 8);
  }
}

// y[b][p][c] = x * s[b] (DropPath) or x * s[b][c] (Dropout2d), optionally + res (the identity branch)
template <typename T>
__global__ void scale_kernel(const T* __restrict__ x, const float* __restrict__ s, const T* __restrict__ res, T* __restrict__ y, int B, long HW,
                             int C, int per_channel) {
  const int c8n = C >> 3;
  const long total = (long)B * HW * c8n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = i % c8n;
    const long p = i / c8n;
    const int b = p / HW;
    float f[8], r[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(x + p * C + c8 * 8), f);
    if (res) unpack8<T>(*reinterpret_cast<const uint4*>(res + p * C + c8 * 8), r);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sc = per_channel ? s[(long)b * C + c8 * 8 + j] : s[b];
      f[j] = f[j] * sc + (res ? r[j] : 0.f);
    }
    *reinterpret_cast<uint4*>(y + p * C + c8 * 8) = pack8<T>(f);
  }
}

// fp32 [n] -> T [n] (the dk / dv accumulators)
// ---- attention backward for T_kv <= 64 on the matrix cores -------------------------------------------------------------------------------
// EfficientMultiheadAttention after the spatial reduction has 64 keys at every stage of a 256 x 256 input (backbone.py:293-313) against
// 64 ... 4096 queries.  The scalar kernel above spends 14 K FMA per query on the vector unit at one wave per workgroup; here a WAVE
// owns a 64-query tile and every product is a v_mfma_f32_16x16x32 (head dim 32 = one k-step):
//   S^T = K Q^T, dP^T = V dO^T        operands straight from global memory (16-byte rows of K / V as A, of Q / dO as B)
//   P = exp(S scale - lse), dS = P (dP - D) scale, D = rowsum(dO . O)     in the accumulator layout (4 consecutive keys x 1 query)
//   dQ^T = K^T dS^T, dK^T += Q^T dS, dV^T += dO^T P        second stage: P / dS in 16 bit through LDS in the two orientations the
//                                                           operands need, Q^T / dO^T / K^T as transposed LDS copies
// dK / dV stay in accumulator registers over all tiles of the wave, the four waves of a block are summed through LDS and ONE wave issues
// the fp32 atomics (the scalar kernel's bottleneck at T_q = 4096: 64 workgroups per image hitting the same 4096 addresses).
// LDS rows are 72 elements (144 B) apart: the 16 rows of a ds_read_b128 group start in 16 different bank quads.
constexpr int KV64_LD = 72;
constexpr int KV64_WAVE_ELEMS = (3 * 64 + 2 * 32) * KV64_LD;      // dS [q][k], dS^T [k][q], P^T [k][q], Q^T [c][q], dO^T [c][q]

template <typename T>
__global__ __launch_bounds__(256) void attn_kv64_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                            const T* __restrict__ o, const T* __restrict__ dout, const float* __restrict__ lse,
                                                            T* __restrict__ dq, float* __restrict__ dkf, float* __restrict__ dvf, int Tq, int Tkv,
                                                            int heads, long ldq, long ldkv, long ldo, float scale, int tiles_per_wave) {
  __shared__ __attribute__((aligned(16))) T sm[4 * KV64_WAVE_ELEMS + 32 * KV64_LD];
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  T* sKt = sm + 4 * KV64_WAVE_ELEMS;                 // [32 c][64 k], shared by the block
  T* sdS = sm + wave * KV64_WAVE_ELEMS;              // [64 q][64 k]
  T* sdSt = sdS + 64 * KV64_LD;                      // [64 k][64 q]
  T* sPt = sdSt + 64 * KV64_LD;                      // [64 k][64 q]
  T* sQt = sPt + 64 * KV64_LD;                       // [32 c][64 q]
  T* sdOt = sQt + 32 * KV64_LD;                      // [32 c][64 q]
  const T* kb = k + (long)b * Tkv * ldkv + h * 32;
  const T* vb = v + (long)b * Tkv * ldkv + h * 32;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  {   // K^T for the whole block: thread = (key, 8-channel chunk)
    const int key = tid >> 2, c0 = (tid & 3) * 8;
    float f[8];
    unpack8<T>(key < Tkv ? *reinterpret_cast<const uint4*>(kb + (long)key * ldkv + c0) : zero4, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) sKt[(c0 + e) * KV64_LD + key] = from_f32<T>(f[e]);
  }
  // A operands of the first stage: rows of K and V (key l15 + 16 kt, channels kg * 8 .. + 7), kept for every tile of this wave
  uint4 fk[4], fv[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = kt * 16 + l15;
    fk[kt] = key < Tkv ? *reinterpret_cast<const uint4*>(kb + (long)key * ldkv + kg * 8) : zero4;
    fv[kt] = key < Tkv ? *reinterpret_cast<const uint4*>(vb + (long)key * ldkv + kg * 8) : zero4;
  }
  f32x4 dKa[2][4], dVa[2][4];      // dK^T / dV^T: [channel tile][key tile], rows = 4 consecutive channels, column = key
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) dKa[ct][kt] = dVa[ct][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int ntiles = (Tq + 63) / 64;
  for (int it = 0; it < tiles_per_wave; ++it) {        // the same trip count for every wave: the barriers below are block-wide
    const int tile = (blockIdx.x * 4 + wave) * tiles_per_wave + it;
    const int q0 = tile * 64;
    const bool live = tile < ntiles;      // wave-uniform: a wave without a tile only keeps the block's barriers company
    // ---- this tile's Q / dO rows (B operands), D and lse per query, transposed copies --------------------------------------------
    uint4 fq[4], fdo[4];
    float Dq[4], Lq[4];
    if (live) {
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int qi = q0 + qt * 16 + l15;
      const bool ok = live && qi < Tq;
      const long row = ((long)b * Tq + (ok ? qi : 0));
      fq[qt] = ok ? *reinterpret_cast<const uint4*>(q + row * ldq + h * 32 + kg * 8) : zero4;
      fdo[qt] = ok ? *reinterpret_cast<const uint4*>(dout + row * ldo + h * 32 + kg * 8) : zero4;
      const uint4 fo = ok ? *reinterpret_cast<const uint4*>(o + row * ldo + h * 32 + kg * 8) : zero4;
      float a[8], d8[8], o8[8];
      unpack8<T>(fq[qt], a);
      unpack8<T>(fdo[qt], d8);
      unpack8<T>(fo, o8);
      float part = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        part += d8[e] * o8[e];
        sQt[(kg * 8 + e) * KV64_LD + qt * 16 + l15] = from_f32<T>(a[e]);
        sdOt[(kg * 8 + e) * KV64_LD + qt * 16 + l15] = from_f32<T>(d8[e]);
      }
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      Dq[qt] = part;
      Lq[qt] = ok ? lse[((long)b * heads + h) * Tq + qi] : 0.f;
    }
    // ---- first stage: S^T and dP^T tiles, P / dS to LDS ---------------------------------------------------------------------------
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 st = Mfma<T>::run(fk[kt], fq[qt], z);       // rows: keys kt*16 + 4*kg + i, column: query qt*16 + l15
        const f32x4 dpt = Mfma<T>::run(fv[kt], fdo[qt], z);
        const int qloc = qt * 16 + l15;
        const bool qok = live && q0 + qloc < Tq;
        float pv[4], dsv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key = kt * 16 + kg * 4 + i;
          const float p = (qok && key < Tkv) ? expf(st[i] * scale - Lq[qt]) : 0.f;
          pv[i] = p;
          dsv[i] = p * (dpt[i] - Dq[qt]) * scale;
          sPt[key * KV64_LD + qloc] = from_f32<T>(p);
          sdSt[key * KV64_LD + qloc] = from_f32<T>(dsv[i]);
        }
        *reinterpret_cast<uint2*>(sdS + qloc * KV64_LD + kt * 16 + kg * 4) = pack4<T>(dsv[0], dsv[1], dsv[2], dsv[3]);
      }
    }
    }
    __syncthreads();
    if (live) {
    // ---- second stage ----------------------------------------------------------------------------------------------------------------
    // dQ^T [c][q] = sum_k K^T[c][k] dS[q][k]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      uint4 ka[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) ka[ks] = *reinterpret_cast<const uint4*>(sKt + (ct * 16 + l15) * KV64_LD + ks * 32 + kg * 8);
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc = Mfma<T>::run(ka[ks], *reinterpret_cast<const uint4*>(sdS + (qt * 16 + l15) * KV64_LD + ks * 32 + kg * 8), acc);
        const int qi = q0 + qt * 16 + l15;
        if (live && qi < Tq)
          *reinterpret_cast<uint2*>(dq + ((long)b * Tq + qi) * ldq + h * 32 + ct * 16 + kg * 4) = pack4<T>(acc[0], acc[1], acc[2], acc[3]);
      }
    }
    // dK^T [c][k] += sum_q Q^T[c][q] dS^T[k][q];   dV^T [c][k] += sum_q dO^T[c][q] P^T[k][q]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      uint4 qa[2], da[2];
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        qa[qs] = *reinterpret_cast<const uint4*>(sQt + (ct * 16 + l15) * KV64_LD + qs * 32 + kg * 8);
        da[qs] = *reinterpret_cast<const uint4*>(sdOt + (ct * 16 + l15) * KV64_LD + qs * 32 + kg * 8);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
          dKa[ct][kt] = Mfma<T>::run(qa[qs], *reinterpret_cast<const uint4*>(sdSt + (kt * 16 + l15) * KV64_LD + qs * 32 + kg * 8), dKa[ct][kt]);
          dVa[ct][kt] = Mfma<T>::run(da[qs], *reinterpret_cast<const uint4*>(sPt + (kt * 16 + l15) * KV64_LD + qs * 32 + kg * 8), dVa[ct][kt]);
        }
      }
    }
    }
    __syncthreads();      // the next tile overwrites this wave's LDS tiles
  }
  // ---- block reduction of dK / dV: waves 1..3 park their accumulators in (their own, now free) LDS region, wave 0 adds and issues atomics
  float* red = reinterpret_cast<float*>(sm + wave * KV64_WAVE_ELEMS);      // [2][2 ct][4 kt][64 lanes][4]: 16 KB of the wave's 36 KB
  if (wave != 0) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        *reinterpret_cast<f32x4*>(red + (((0 * 2 + ct) * 4 + kt) * 64 + lane) * 4) = dKa[ct][kt];
        *reinterpret_cast<f32x4*>(red + (((1 * 2 + ct) * 4 + kt) * 64 + lane) * 4) = dVa[ct][kt];
      }
  }
  __syncthreads();
  if (wave == 0) {
    const int H32 = heads * 32;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 sk = dKa[ct][kt], sv = dVa[ct][kt];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float* rw = reinterpret_cast<const float*>(sm + w * KV64_WAVE_ELEMS);
          const f32x4 a = *reinterpret_cast<const f32x4*>(rw + (((0 * 2 + ct) * 4 + kt) * 64 + lane) * 4);
          const f32x4 c2 = *reinterpret_cast<const f32x4*>(rw + (((1 * 2 + ct) * 4 + kt) * 64 + lane) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sk[i] += a[i];
            sv[i] += c2[i];
          }
        }
        const int key = kt * 16 + l15;
        if (key < Tkv) {
          float* dkp = dkf + ((long)b * Tkv + key) * H32 + h * 32 + ct * 16 + kg * 4;
          float* dvp = dvf + ((long)b * Tkv + key) * H32 + h * 32 + ct * 16 + kg * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            atomicAdd(dkp + i, sk[i]);
            atomicAdd(dvp + i, sv[i]);
          }
        }
      }
  }
}

// forward of the same attention on the matrix cores: S^T = K Q^T per 64-query tile of a wave, softmax over the 64 keys of a query column
// (16 values per lane, the rest two xor-shuffles away), P in 16 bit through LDS, O^T = V^T P^T
template <typename T>
__global__ __launch_bounds__(256) void attn_kv64_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                            T* __restrict__ o, float* __restrict__ lse, int Tq, int Tkv, int heads, long ldq,
                                                            long ldkv, long ldo, float scale, int tiles_per_wave) {
  __shared__ __attribute__((aligned(16))) T sm[4 * 64 * KV64_LD + 32 * KV64_LD];
  const int h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  T* sP = sm + wave * 64 * KV64_LD;                  // [64 q][64 k] of this wave
  T* sVt = sm + 4 * 64 * KV64_LD;                    // [32 c][64 k], shared by the block
  const T* kb = k + (long)b * Tkv * ldkv + h * 32;
  const T* vb = v + (long)b * Tkv * ldkv + h * 32;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  {
    const int key = tid >> 2, c0 = (tid & 3) * 8;
    float f[8];
    unpack8<T>(key < Tkv ? *reinterpret_cast<const uint4*>(vb + (long)key * ldkv + c0) : zero4, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) sVt[(c0 + e) * KV64_LD + key] = from_f32<T>(f[e]);
  }
  uint4 fk[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int key = kt * 16 + l15;
    fk[kt] = key < Tkv ? *reinterpret_cast<const uint4*>(kb + (long)key * ldkv + kg * 8) : zero4;
  }
  __syncthreads();
  const int ntiles = (Tq + 63) / 64;
  for (int it = 0; it < tiles_per_wave; ++it) {
    const int tile = (blockIdx.x * 4 + wave) * tiles_per_wave + it;
    const int q0 = tile * 64;
    const bool live = tile < ntiles;
    float linv[4];
    if (live) {
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int qi = q0 + qt * 16 + l15;
      const bool ok = live && qi < Tq;
      const uint4 fq = ok ? *reinterpret_cast<const uint4*>(q + ((long)b * Tq + qi) * ldq + h * 32 + kg * 8) : zero4;
      f32x4 st[4];
      float m = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        st[kt] = Mfma<T>::run(fk[kt], fq, (f32x4){0.f, 0.f, 0.f, 0.f});      // rows: keys kt*16 + 4*kg + i, column: query qt*16 + l15
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          st[kt][i] = (kt * 16 + kg * 4 + i < Tkv) ? st[kt][i] * scale : -3.0e38f;
          m = fmaxf(m, st[kt][i]);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        float pv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pv[i] = (kt * 16 + kg * 4 + i < Tkv) ? expf(st[kt][i] - m) : 0.f;
          l += pv[i];
        }
        *reinterpret_cast<uint2*>(sP + (qt * 16 + l15) * KV64_LD + kt * 16 + kg * 4) = pack4<T>(pv[0], pv[1], pv[2], pv[3]);
      }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      linv[qt] = 1.0f / l;
      if (ok && lse && kg == 0) lse[((long)b * heads + h) * Tq + qi] = m + logf(l);
    }
    }
    __syncthreads();
    if (live) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      uint4 va[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) va[ks] = *reinterpret_cast<const uint4*>(sVt + (ct * 16 + l15) * KV64_LD + ks * 32 + kg * 8);
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          acc = Mfma<T>::run(va[ks], *reinterpret_cast<const uint4*>(sP + (qt * 16 + l15) * KV64_LD + ks * 32 + kg * 8), acc);
        const int qi = q0 + qt * 16 + l15;
        if (live && qi < Tq)
          *reinterpret_cast<uint2*>(o + ((long)b * Tq + qi) * ldo + h * 32 + ct * 16 + kg * 4) =
              pack4<T>(acc[0] * linv[qt], acc[1] * linv[qt], acc[2] * linv[qt], acc[3] * linv[qt]);
      }
    }
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void f32_to_t_kernel(const float* __restrict__ x, T* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = from_f32<T>(x[i]);
}
// (dk, dv) fp32 [rows][C] each -> 16-bit rows of pitch ldo (dk) / ldo (dv): one launch for both, straight into a [rows][2 C] buffer
template <typename T>
__global__ void f32x2_to_t_kernel(const float* __restrict__ a, const float* __restrict__ b, T* __restrict__ ya, T* __restrict__ yb, long rows, int C,
                                  long ldo) {
  const long n = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < 2 * n; i += (long)gridDim.x * blockDim.x) {
    const bool second = i >= n;
    const long j = second ? i - n : i;
    const long r = j / C;
    const int c = (int)(j - r * C);
    (second ? yb : ya)[r * ldo + c] = from_f32<T>((second ? b : a)[j]);
  }
}

int lanes_per_row(int C) {
  int chunks = C / 8, l = 1;
  while (l < chunks) l <<= 1;
  return l;
}

}  // namespace

extern "C" int jg_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mr, int64_t R, int C, float eps,
                                jg_stream_t s) {
  if (!x || !gamma || !beta || !y || R < 1 || C < 8 || C % 8 || C > 512) return JG_ERR_BAD_ARG;
  const int lpr = lanes_per_row(C);
  const long waves = (R + 64 / lpr - 1) / (64 / lpr);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((layernorm_fwd_kernel<T>), dim3(grid_for(waves, 4, 8192)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                              gamma, beta, (T*)y, mr, (long)R, C, lpr, eps););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_layernorm_fwd_add(int dtype, const void* x, const void* add, const float* scale, int64_t rows_per_image, void* xsum,
                                    const float* gamma, const float* beta, void* y, float* mr, int64_t R, int C, float eps, jg_stream_t s) {
  if (!x || !add || !xsum || !gamma || !beta || !y || R < 1 || C < 8 || C % 8 || C > 512 || rows_per_image < 1) return JG_ERR_BAD_ARG;
  const int lpr = lanes_per_row(C);
  const long waves = (R + 64 / lpr - 1) / (64 / lpr);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((layernorm_fwd_kernel<T>), dim3(grid_for(waves, 4, 8192)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                              gamma, beta, (T*)y, mr, (long)R, C, lpr, eps, (const T*)add, scale, (long)rows_per_image, (T*)xsum););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_layernorm_bwd_add(int dtype, const void* x, const void* dy, const float* gamma, const float* mr, const void* res, void* dx,
                                    float* dgamma, float* dbeta, int64_t R, int C, jg_stream_t s) {
  if (!x || !dy || !gamma || !mr || R < 1 || C < 8 || C % 8 || C > 512 || (!dgamma != !dbeta) || (res && !dx)) return JG_ERR_BAD_ARG;
  // (with the parameter gradients every block ends with 2 C global atomics onto the same addresses: 256 blocks instead of 1024 keep
  //  that chain short -- it, not the streaming, set the 20 us floor of this launch)
  const int lpr = lanes_per_row(C);
  const long waves = (R + 64 / lpr - 1) / (64 / lpr);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((layernorm_bwd_kernel<T>), dim3(grid_for(waves, 4, dgamma ? jg_tune(JG_TUNE_LN_BWD_CAP) : 1024)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                              (const T*)dy, gamma, mr, (const T*)res, (T*)dx, dgamma, dbeta, (long)R, C, lpr););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_layernorm_bwd_add2(int dtype, const void* x, const void* dy, const float* gamma, const float* mr, const void* res, void* dx,
                                     const float* scale, int64_t rows_per_image, void* dx2, float* dgamma, float* dbeta, int64_t R, int C, jg_stream_t s) {
  if (!x || !dy || !gamma || !mr || !dx || !dx2 || R < 1 || C < 8 || C % 8 || C > 512 || (!dgamma != !dbeta) || rows_per_image < 1) return JG_ERR_BAD_ARG;
  const int lpr = lanes_per_row(C);
  const long waves = (R + 64 / lpr - 1) / (64 / lpr);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((layernorm_bwd_kernel<T>), dim3(grid_for(waves, 4, dgamma ? jg_tune(JG_TUNE_LN_BWD_CAP) : 1024)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                              (const T*)dy, gamma, mr, (const T*)res, (T*)dx, dgamma, dbeta, (long)R, C, lpr, scale, (long)rows_per_image, (T*)dx2););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* mr, void* dx, float* dgamma,
                                float* dbeta, int64_t R, int C, jg_stream_t s) {
  return jg_layernorm_bwd_add(dtype, x, dy, gamma, mr, nullptr, dx, dgamma, dbeta, R, C, s);
}
extern "C" int jg_dwconv3x3_fwd_pad(int dtype, const void* x, const float* w, const float* bias, void* pre, void* y, int B, int H, int W, int C,
                                    int gelu, int pad_mode, jg_stream_t s) {
  if (!x || !w || !y || B < 1 || H < 1 || W < 1 || C < 8 || C % 8) return JG_ERR_BAD_ARG;
  if (C > 2048) return JG_ERR_UNSUPPORTED;
  const int tpb_f = 256 / (C / 8) > 0 ? 256 / (C / 8) : 1;
  if (pad_mode != 0 && pad_mode != 1) return JG_ERR_BAD_ARG;
  if (pad_mode == 1 && (H < 2 || W < 2)) return JG_ERR_BAD_ARG;
  if (jg_tune(JG_TUNE_DW_RUN) != 0 && W >= 4) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv3x3_fwd_run_kernel<T>), dim3(grid_for((long)B * H * ((W + DW_RX - 1) / DW_RX), tpb_f, 8192)), dim3(256), 0,
                                                (hipStream_t)s, (const T*)x, w, bias, (T*)pre, (T*)y, B, H, W, C, gelu, pad_mode););
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T>), dim3(grid_for((long)B * H * W, tpb_f * 4, 4096)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)x, w, bias, (T*)pre, (T*)y, B, H, W, C, gelu, pad_mode););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_dwconv3x3_fwd(int dtype, const void* x, const float* w, const float* bias, void* pre, void* y, int B, int H, int W, int C,
                                int gelu, jg_stream_t s) {
  return jg_dwconv3x3_fwd_pad(dtype, x, w, bias, pre, y, B, H, W, C, gelu, 0, s);
}
static int dw_bwd_blocks(long npix, int C) {
  const int c8n = C / 8;
  const int tpb = 256 / c8n > 0 ? 256 / c8n : 1;
  const int ppt = jg_tune(JG_TUNE_DW_BWD_PPT) > 0 ? jg_tune(JG_TUNE_DW_BWD_PPT) : 8;   // pixels per thread
  return grid_for(npix, tpb * ppt, jg_tune(JG_TUNE_DW_BWD_CAP));
}
extern "C" int64_t jg_dwconv3x3_bwd_ws_floats(int B, int H, int W, int C) {
  if (B < 1 || H < 1 || W < 1 || C < 8 || C % 8) return 0;
  return (int64_t)dw_bwd_blocks((long)B * H * W, C) * C * 10;
}
extern "C" int jg_dwconv3x3_bwd_ws(int dtype, const void* x, const void* pre, const void* dy, const float* w, void* du, void* dx, float* dw,
                                   float* dbias, float* ws, int64_t ws_floats, int B, int H, int W, int C, int gelu, jg_stream_t s) {
  return jg_dwconv3x3_bwd_ws_pad(dtype, x, pre, dy, w, du, dx, dw, dbias, ws, ws_floats, B, H, W, C, gelu, 0, s);
}
extern "C" int jg_dwconv3x3_bwd_ws_pad(int dtype, const void* x, const void* pre, const void* dy, const float* w, void* du, void* dx, float* dw,
                                       float* dbias, float* ws, int64_t ws_floats, int B, int H, int W, int C, int gelu, int pad_mode, jg_stream_t s) {
  if (!x || !dy || !w || !du || B < 1 || H < 1 || W < 1 || C < 8 || C % 8 || (gelu && !pre)) return JG_ERR_BAD_ARG;
  if ((pad_mode != 0 && pad_mode != 1) || (pad_mode == 1 && (H < 4 || W < 4))) return JG_ERR_BAD_ARG;
  if (C > 2048) return JG_ERR_UNSUPPORTED;   // one 8-channel chunk per thread of a 256-thread block
  hipStream_t st = (hipStream_t)s;
  const int c8n = C / 8;
  const int tpb = 256 / c8n > 0 ? 256 / c8n : 1;
  const long npix = (long)B * H * W;
  const int G = dw_bwd_blocks(npix, C);
  if (ws && ws_floats < (int64_t)G * C * 10) return JG_ERR_BAD_ARG;
  if (!dw && !dbias) ws = nullptr;   // nothing to reduce: the kernel only writes du
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv3x3_bwd_w_kernel<T>), dim3(G), dim3(256), 0, st,
                                              (const T*)x, (const T*)pre, (const T*)dy, (T*)du, dw, dbias, ws, B, H, W, C, gelu, pad_mode););
  if (ws) {
    hipLaunchKernelGGL(dw_partials_sum_kernel, dim3((C * 10 + 63) / 64, (G + 63) / 64), dim3(256), 0, st, (const float*)ws, G, C * 10, c8n, dw, dbias);
  }
  if (dx) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv3x3_bwd_x_kernel<T>), dim3(grid_for(npix, tpb * 4, 4096)), dim3(256), 0, st, (const T*)du, w,
                                                (T*)dx, B, H, W, C, pad_mode););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_dwconv3x3_bwd(int dtype, const void* x, const void* pre, const void* dy, const float* w, void* du, void* dx, float* dw,
                                float* dbias, int B, int H, int W, int C, int gelu, jg_stream_t s) {
  return jg_dwconv3x3_bwd_ws(dtype, x, pre, dy, w, du, dx, dw, dbias, nullptr, 0, B, H, W, C, gelu, s);
}
extern "C" int jg_attn_smallkv_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq, int Tkv, int heads,
                                   int64_t ldq, int64_t ldkv, int64_t ldo, float scale, jg_stream_t s) {
  if (!q || !k || !v || !o || B < 1 || Tq < 1 || Tkv < 1 || heads < 1 || ldq % 8 || ldkv % 8 || ldo % 8) return JG_ERR_BAD_ARG;
  if (Tkv > AKV_MAX || B > 65535 || heads > 65535) return JG_ERR_UNSUPPORTED;
  const dim3 grid((Tq + 63) / 64, heads, B);
  if (Tkv <= 64) {
    const int ntiles = (Tq + 63) / 64;
    int tpw = 1;
    while (tpw < 8 && 4 * tpw * 2 <= ntiles && (long)((ntiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) * heads * B >= 512) tpw *= 2;   // (never more tiles per block than exist: at T_q = 64, batch 64, 8 heads every wave ran 8 trips for ONE live tile)
    const dim3 gridm((ntiles + 4 * tpw - 1) / (4 * tpw), heads, B);
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_kv64_fwd_kernel<T>), gridm, dim3(256), 0, (hipStream_t)s, (const T*)q, (const T*)k,
                                                (const T*)v, (T*)o, lse, Tq, Tkv, heads, (long)ldq, (long)ldkv, (long)ldo, scale, tpw););
  } else {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_smallkv_fwd_kernel<T, AKV_MAX>), grid, dim3(64), 0, (hipStream_t)s, (const T*)q, (const T*)k,
                                                (const T*)v, (T*)o, lse, Tq, Tkv, heads, (long)ldq, (long)ldkv, (long)ldo, scale););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_attn_smallkv_bwd2(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                    void* dq, float* dkf, float* dvf, void* dk, void* dv, int64_t lddkv, int B, int Tq, int Tkv, int heads,
                                    int64_t ldq, int64_t ldkv, int64_t ldo, float scale, jg_stream_t s) {
  if (!q || !k || !v || !o || !dout || !lse || !dq || !dkf || !dvf || !dk || !dv || lddkv < (int64_t)heads * 32) return JG_ERR_BAD_ARG;
  if (Tkv > AKV_MAX || B > 65535 || heads > 65535) return JG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)s;
  const long nkv = (long)B * Tkv * heads * 32;
  if (dvf == dkf + nkv) {          // one contiguous accumulator pair: one fill
    if (hipMemsetAsync(dkf, 0, 2 * nkv * sizeof(float), st) != hipSuccess) return JG_ERR_LAUNCH;
  } else if (hipMemsetAsync(dkf, 0, nkv * sizeof(float), st) != hipSuccess || hipMemsetAsync(dvf, 0, nkv * sizeof(float), st) != hipSuccess) {
    return JG_ERR_LAUNCH;
  }
  const dim3 grid((Tq + 63) / 64, heads, B);
  if (Tkv <= 64 && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0) {
    // matrix-core kernel: a wave per 64-query tile, 4 waves per block; tiles per wave chosen so that the grid still has >= 512 blocks
    const int ntiles = (Tq + 63) / 64;
    int tpw = 1;
    while (tpw < 8 && 4 * tpw * 2 <= ntiles && (long)((ntiles + 4 * tpw * 2 - 1) / (4 * tpw * 2)) * heads * B >= 512) tpw *= 2;   // (never more tiles per block than exist: at T_q = 64, batch 64, 8 heads every wave ran 8 trips for ONE live tile)
    const dim3 gridm((ntiles + 4 * tpw - 1) / (4 * tpw), heads, B);
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_kv64_bwd_kernel<T>), gridm, dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v,
                                                (const T*)o, (const T*)dout, lse, (T*)dq, dkf, dvf, Tq, Tkv, heads, (long)ldq, (long)ldkv, (long)ldo,
                                                scale, tpw););
  } else if (Tkv <= 64) {
    const dim3 grid4((Tq + 63) / 64, heads, B);       // QT = 1 query tile per block
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_smallkv_bwd_kernel<T, 64>), grid4, dim3(64), 0, st, (const T*)q, (const T*)k, (const T*)v,
                                                (const T*)o, (const T*)dout, lse, (T*)dq, dkf, dvf, Tq, Tkv, heads, (long)ldq, (long)ldkv, (long)ldo,
                                                scale););
  } else {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_smallkv_bwd_kernel<T, AKV_MAX>), grid, dim3(64), 0, st, (const T*)q, (const T*)k, (const T*)v,
                                                (const T*)o, (const T*)dout, lse, (T*)dq, dkf, dvf, Tq, Tkv, heads, (long)ldq, (long)ldkv, (long)ldo,
                                                scale););
  }
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((f32x2_to_t_kernel<T>), dim3(grid_for(2 * nkv)), dim3(256), 0, st, dkf, dvf, (T*)dk, (T*)dv,
                                              (long)B * Tkv, heads * 32, (long)lddkv););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_attn_smallkv_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                   void* dq, float* dkf, float* dvf, void* dk, void* dv, int B, int Tq, int Tkv, int heads, int64_t ldq,
                                   int64_t ldkv, int64_t ldo, float scale, jg_stream_t s) {
  return jg_attn_smallkv_bwd2(dtype, q, k, v, o, dout, lse, dq, dkf, dvf, dk, dv, (int64_t)heads * 32, B, Tq, Tkv, heads, ldq, ldkv, ldo, scale, s);
}
extern "C" int jg_bilinear2_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int64_t ldy, int align_corners,
                                jg_stream_t s) {
  if (!x || !y || C < 8 || C % 8 || ldy < C || ldy % 8 || H < 1 || W < 1 || Ho < 1 || Wo < 1) return JG_ERR_BAD_ARG;
  if (align_corners) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bilinear_fwd_kernel<T, true>), dim3(grid_for((long)B * Ho * Wo * (C / 8))), dim3(256), 0,
                                                (hipStream_t)s, (const T*)x, (T*)y, B, H, W, C, Ho, Wo, (long)ldy););
  } else {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bilinear_fwd_kernel<T, false>), dim3(grid_for((long)B * Ho * Wo * (C / 8))), dim3(256), 0,
                                                (hipStream_t)s, (const T*)x, (T*)y, B, H, W, C, Ho, Wo, (long)ldy););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_bilinear_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int64_t ldy, jg_stream_t s) {
  return jg_bilinear2_fwd(dtype, x, y, B, H, W, C, Ho, Wo, ldy, 0, s);
}
extern "C" int jg_bilinear2_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo, int64_t lddy, int align_corners,
                                jg_stream_t s) {
  if (!dy || !dx || C < 8 || C % 8 || lddy < C || lddy % 8 || Ho < H || Wo < W) return JG_ERR_BAD_ARG;   // up-sampling only
  if (align_corners) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bilinear_bwd_kernel<T, true>), dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0,
                                                (hipStream_t)s, (const T*)dy, (T*)dx, B, H, W, C, Ho, Wo, (long)lddy););
  } else {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bilinear_bwd_kernel<T, false>), dim3(grid_for((long)B * H * W * (C / 8))), dim3(256), 0,
                                                (hipStream_t)s, (const T*)dy, (T*)dx, B, H, W, C, Ho, Wo, (long)lddy););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_resize_sum(int dtype, const void* x0, const void* x1, int H1, int W1, const void* x2, int H2, int W2, const void* x3, int H3, int W3,
                             void* y, int B, int Ho, int Wo, int C, int act, const float* chscale, jg_stream_t s) {
  if (!x0 || !y || B < 1 || Ho < 1 || Wo < 1 || C < 8 || C % 8 || (act != JG_ACT_NONE && act != JG_ACT_RELU)) return JG_ERR_BAD_ARG;
  ResizeSumP rp;
  const void* xs[3] = {x1, x2, x3};
  const int Hs[3] = {H1, H2, H3}, Ws[3] = {W1, W2, W3};
  rp.n = 0;
  for (int t = 0; t < 3; ++t) {
    rp.x[t] = nullptr; rp.H[t] = 1; rp.W[t] = 1;
  }
  for (int t = 0; t < 3; ++t) {
    if (!xs[t]) continue;
    if (Hs[t] < 1 || Ws[t] < 1) return JG_ERR_BAD_ARG;
    rp.x[rp.n] = xs[t]; rp.H[rp.n] = Hs[t]; rp.W[rp.n] = Ws[t];
    ++rp.n;
  }
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_sum_kernel<T>), dim3(grid_for((long)B * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)s, (const T*)x0, rp,
                                              (T*)y, B, C, Ho, Wo, act, chscale););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo, int64_t lddy, jg_stream_t s) {
  return jg_bilinear2_bwd(dtype, dy, dx, B, H, W, C, Ho, Wo, lddy, 0, s);
}
// workspace of jg_resize_sum_bwd in floats: B * Ho * (W1 + W2 + W3) * C over the terms that are present
extern "C" int64_t jg_resize_sum_bwd_ws_floats(int B, int Ho, int C, int W1, int W2, int W3) {
  return (int64_t)B * Ho * C * ((W1 > 0 ? W1 : 0) + (W2 > 0 ? W2 : 0) + (W3 > 0 ? W3 : 0));
}
extern "C" int jg_resize_sum_bwd(int dtype, const void* y, const void* dy, void* g, void* dx1, int H1, int W1, void* dx2, int H2, int W2, void* dx3,
                                 int H3, int W3, float* ws, int B, int Ho, int Wo, int C, int act, const float* chscale, jg_stream_t s) {
  if (!dy || B < 1 || Ho < 1 || Wo < 1 || C < 8 || C % 8 || (act != JG_ACT_NONE && act != JG_ACT_RELU) || (act == JG_ACT_RELU && !y)) return JG_ERR_BAD_ARG;
  const size_t shm = (size_t)Wo * C * 2;
  if (shm > 65536) return JG_ERR_UNSUPPORTED;        // the row of g in LDS: the caller keeps the gather kernels for wider rows
  ResizeBwdP rp;
  void* dxs[3] = {dx1, dx2, dx3};
  const int Hs[3] = {H1, H2, H3}, Ws[3] = {W1, W2, W3};
  rp.n = 0;
  long off = 0, threads = 0;
  for (int t = 0; t < 3; ++t) {
    rp.dx[t] = nullptr; rp.H[t] = 1; rp.W[t] = 1; rp.off[t] = 0;
  }
  for (int t = 0; t < 3; ++t) {
    if (!dxs[t]) continue;
    if (Hs[t] < 1 || Ws[t] < 1 || Hs[t] > Ho || Ws[t] > Wo) return JG_ERR_BAD_ARG;
    rp.dx[rp.n] = dxs[t]; rp.H[rp.n] = Hs[t]; rp.W[rp.n] = Ws[t]; rp.off[rp.n] = off;
    off += (long)B * Ho * Ws[t] * C;
    threads += (long)B * Hs[t] * Ws[t] * (C / 8);
    ++rp.n;
  }
  if (rp.n && !ws) return JG_ERR_BAD_ARG;
  if (!rp.n && !g) return JG_OK;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_sum_bwd_x_kernel<T>), dim3((unsigned)((long)B * Ho)), dim3(256), shm, (hipStream_t)s, (const T*)y,
                                              (const T*)dy, (T*)g, rp, ws, C, Ho, Wo, act, chscale););
  if (rp.n) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((resize_sum_bwd_y_kernel<T>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)s, ws, rp, B,
                                                C, Ho););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_bn_coef(const float* sums, const float* gamma, const float* beta, float* running_mean, float* running_var, float* ab, float* mr,
                          int B, int HW, int C, float eps, float momentum, int training, jg_stream_t s) {
  if (!gamma || !beta || !ab || !mr || B < 1 || HW < 1 || C < 1 || (training && !sums) || (!training && (!running_mean || !running_var)))
    return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, sums, gamma, beta, running_mean, running_var, ab, mr, B, HW, C, eps,
                     momentum, training);
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_bn_bwd_coef(const float* red, const float* gamma, const float* mr, float* pqr, float* dgamma, float* dbeta, int B, int HW, int C,
                              int training, jg_stream_t s) {
  if (!red || !gamma || !mr || !pqr || B < 1 || HW < 1 || C < 1) return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, red, gamma, mr, pqr, dgamma, dbeta, B, HW, C, training);
  JG_CHECK_LAUNCH();
  return JG_OK;
}
static bool acomp_px_ok(int f, int na, int nc, int ldimg, int ldl, int ldx, int ldo, long npix) {
  const bool fok = f == 1 || f == 2 || f == 4 || f == 8;
  return fok && na <= 16 && nc <= 8 && ldimg % 8 == 0 && ldimg <= 64 && ldl % 8 == 0 && ldl <= 16 && ldx % 8 == 0 && ldo % 8 == 0 &&
         npix < (1L << 31) * 256;
}
extern "C" int jg_attn_compose_fwd(int dtype, const void* img, const void* logits, const void* xin, void* out, int B, int S, int f, int na, int ni,
                                   int nc, int ldimg, int ldl, int ldx, int ldo, jg_stream_t s) {
  if (!img || !logits || !xin || !out || B < 1 || S < 1 || f < 1 || S % f || na < 1 || na > ACOMP_MAX || ni < 0 || ni > na || nc < 1) return JG_ERR_BAD_ARG;
  if (ldimg < nc * ni || ldl < na || ldx < nc || ldo < nc) return JG_ERR_BAD_ARG;
  if (acomp_px_ok(f, na, nc, ldimg, ldl, ldx, ldo, (long)B * S * S)) {
    const long total = (long)B * S * S;
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_compose_fwd_px_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                                                (const T*)img, (const T*)logits, (const T*)xin, (T*)out, B, S, f, na, ni, nc, ldimg, ldl, ldx, ldo););
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_compose_fwd_kernel<T>), dim3(grid_for((long)B * (S / f) * (S / f), 64)), dim3(64), 0, (hipStream_t)s,
                                              (const T*)img, (const T*)logits, (const T*)xin, (T*)out, B, S, f, na, ni, nc, ldimg, ldl, ldx, ldo););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_attn_compose_bwd(int dtype, const void* img, const void* logits, const void* xin, const void* dout, void* dimg, void* dlogits,
                                   void* dxin, int B, int S, int f, int na, int ni, int nc, int ldimg, int ldl, int ldx, int ldo, jg_stream_t s) {
  if (!img || !logits || !xin || !dout || !dimg || !dlogits || B < 1 || S < 1 || f < 1 || S % f || na < 1 || na > ACOMP_MAX || ni < 0 || ni > na)
    return JG_ERR_BAD_ARG;
  if (acomp_px_ok(f, na, nc, ldimg, ldl, ldx, ldo, (long)B * S * S)) {
    const long total = (long)B * S * S;
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_compose_bwd_px_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s,
                                                (const T*)img, (const T*)logits, (const T*)xin, (const T*)dout, (T*)dimg, (T*)dlogits, (T*)dxin, B, S,
                                                f, na, ni, nc, ldimg, ldl, ldx, ldo););
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_compose_bwd_kernel<T>), dim3(grid_for((long)B * (S / f) * (S / f), 64)), dim3(64), 0, (hipStream_t)s,
                                              (const T*)img, (const T*)logits, (const T*)xin, (const T*)dout, (T*)dimg, (T*)dlogits, (T*)dxin, B, S, f,
                                              na, ni, nc, ldimg, ldl, ldx, ldo););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_scale(int dtype, const void* x, const float* scale, const void* res, void* y, int B, int64_t HW, int C, int per_channel,
                        jg_stream_t s) {
  if (!x || !scale || !y || B < 1 || HW < 1 || C < 8 || C % 8) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((scale_kernel<T>), dim3(grid_for((long)B * HW * (C / 8))), dim3(256), 0, (hipStream_t)s, (const T*)x, scale,
                                              (const T*)res, (T*)y, B, (long)HW, C, per_channel););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

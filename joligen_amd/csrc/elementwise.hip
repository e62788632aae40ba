// Bandwidth-bound helper kernels of the DDPM step: resampling, channel concat/split, attention
// softmax and head transposes, the small fp32 embedding linears, the DDPM q_sample / loss glue
// and the NCHW<->NHWC converters at the module boundary.  16-byte vector access everywhere the
// layout allows it.
#include "common.h"

namespace {

// ---- 2x2 sum pool / nearest upsample --------------------------------------------------------
template <typename T>
__global__ void pool2x2_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int B, int H, int W, int C,
                               float scale) {
  const int Ho = H / 2, Wo = W / 2, noct = C / 8;
  const long total = (long)B * Ho * Wo * noct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % noct;
    long t = i / noct;
    const int ow = t % Wo; t /= Wo;
    const int oh = t % Ho;
    const int b = t / Ho;
    const T* p00 = x + (((long)b * H + oh * 2) * W + ow * 2) * ldx + co * 8;
    float f[8], a[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(p00), a);
    unpack8<T>(*reinterpret_cast<const uint4*>(p00 + ldx), f);
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] += f[q];
    unpack8<T>(*reinterpret_cast<const uint4*>(p00 + (long)W * ldx), f);
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] += f[q];
    unpack8<T>(*reinterpret_cast<const uint4*>(p00 + (long)W * ldx + ldx), f);
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = (a[q] + f[q]) * scale;
    *reinterpret_cast<uint4*>(y + (((long)b * Ho + oh) * Wo + ow) * ldy + co * 8) = pack8<T>(a);
  }
}

template <typename T>
__global__ void upsample2x_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int B, int H, int W, int C,
                                  float scale) {
  const int Ho = H * 2, Wo = W * 2, noct = C / 8;
  const long total = (long)B * Ho * Wo * noct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % noct;
    long t = i / noct;
    const int ow = t % Wo; t /= Wo;
    const int oh = t % Ho;
    const int b = t / Ho;
    uint4 v = *reinterpret_cast<const uint4*>(x + (((long)b * H + oh / 2) * W + ow / 2) * ldx + co * 8);
    if (scale != 1.0f) {
      float f[8];
      unpack8<T>(v, f);
#pragma unroll
      for (int q = 0; q < 8; ++q) f[q] *= scale;
      v = pack8<T>(f);
    }
    *reinterpret_cast<uint4*>(y + i * 8) = v;
  }
}

template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, long ldsrc, long soff, T* __restrict__ dst, long lddst,
                                     long doff, long P, int n) {
  const int noct = n / 8;
  const long total = P * noct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % noct;
    const long p = i / noct;
    *reinterpret_cast<uint4*>(dst + p * lddst + doff + co * 8) =
        *reinterpret_cast<const uint4*>(src + p * ldsrc + soff + co * 8);
  }
}

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ a, float alpha, const float* __restrict__ alpha_dev,
                             const T* __restrict__ b, float beta, T* __restrict__ y, long n8) {
  if (alpha_dev) alpha *= *alpha_dev;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float fa[8], fb[8];
    unpack8<T>(reinterpret_cast<const uint4*>(a)[i], fa);
    if (b) {
      unpack8<T>(reinterpret_cast<const uint4*>(b)[i], fb);
#pragma unroll
      for (int q = 0; q < 8; ++q) fa[q] = alpha * fa[q] + beta * fb[q];
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) fa[q] *= alpha;
    }
    reinterpret_cast<uint4*>(y)[i] = pack8<T>(fa);
  }
}

// ---- attention helpers -----------------------------------------------------------------------
template <typename T>
__global__ void transpose_heads_kernel(const T* __restrict__ src, long ldsrc, long coff, long hstride,
                                       T* __restrict__ dst, int B, int Tn, int nh, int ch) {
  const int noct = ch / 8;
  const long total = (long)B * nh * noct * Tn;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = i % Tn;  // fastest: 2-byte stores of neighbouring lanes are contiguous
    long r = i / Tn;
    const int co = r % noct; r /= noct;
    const int h = r % nh;
    const int b = r / nh;
    const uint4 v = *reinterpret_cast<const uint4*>(src + ((long)b * Tn + t) * ldsrc + coff + h * hstride + co * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint16_t* d = reinterpret_cast<uint16_t*>(dst) + (((long)b * nh + h) * ch + co * 8) * Tn + t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      d[(long)(2 * q) * Tn] = (uint16_t)(w[q] & 0xffffu);
      d[(long)(2 * q + 1) * Tn] = (uint16_t)(w[q] >> 16);
    }
  }
}

// one wave per row, 4 rows per 256-thread block; T % 4 == 0
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, T* __restrict__ P, long rows, int Tn) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* s4 = reinterpret_cast<const float4*>(S + row * Tn);
  const int n4 = Tn / 4;
  float mx = -INFINITY;
  for (int j = lane; j < n4; j += 64) {
    const float4 v = s4[j];
    mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n4; j += 64) {
    const float4 v = s4[j];
    sum += __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx);
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  uint2* p2 = reinterpret_cast<uint2*>(P + row * Tn);
  for (int j = lane; j < n4; j += 64) {
    const float4 v = s4[j];
    p2[j] = pack4<T>(__expf(v.x - mx) * inv, __expf(v.y - mx) * inv, __expf(v.z - mx) * inv, __expf(v.w - mx) * inv);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, const float* __restrict__ dP,
                                                          T* __restrict__ dS, long rows, int Tn, float alpha) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* g4 = reinterpret_cast<const float4*>(dP + row * Tn);
  const uint2* p2 = reinterpret_cast<const uint2*>(P + row * Tn);
  const int n4 = Tn / 4;
  float dot = 0.f;
  for (int j = lane; j < n4; j += 64) {
    const float4 g = g4[j];
    float pf[4];
    unpack4<T>(p2[j], pf);
    dot += g.x * pf[0] + g.y * pf[1] + g.z * pf[2] + g.w * pf[3];
  }
  dot = wave_sum(dot);
  uint2* o2 = reinterpret_cast<uint2*>(dS + row * Tn);
  for (int j = lane; j < n4; j += 64) {
    const float4 g = g4[j];
    float pf[4];
    unpack4<T>(p2[j], pf);
    o2[j] = pack4<T>(alpha * pf[0] * (g.x - dot), alpha * pf[1] * (g.y - dot), alpha * pf[2] * (g.z - dot),
                     alpha * pf[3] * (g.w - dot));
  }
}

// ---- small fp32 linears (embedding path) --------------------------------------------------------
__device__ __forceinline__ float act_f(float v, int act) {
  return act == JG_ACT_SILU ? v / (1.0f + expf(-v)) : act == JG_ACT_RELU ? fmaxf(v, 0.f) : v;
}
__device__ __forceinline__ float act_grad_f(float v, int act) {
  if (act == JG_ACT_RELU) return v > 0.f ? 1.0f : 0.f;
  if (act != JG_ACT_SILU) return 1.0f;
  const float s = 1.0f / (1.0f + expf(-v));
  return s * (1.0f + v * (1.0f - s));
}

// (few rows, very long reduction: the stacked embedding projection of the UNet, Bn = batch, N = sum of 2C over all ResBlocks)
// dx[b][k] = act'(x[b][k]) * sum_n dy[b][n] W[n][k]: one block per (b, 32-wide k tile); threads stride
// over n (W rows are read as contiguous 128-byte segments), 32 partial sums per thread, LDS reduce.
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ dy, float* __restrict__ dx, int Bn,
                                                            int K, int N, int act) {
  __shared__ float s_red[8][33];
  const int b = blockIdx.x, k0 = blockIdx.y * 32;
  const int kt = min(32, K - k0);
  float acc[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) acc[q] = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float g = dy[(long)b * N + n];
    const float* wr = W + (long)n * K + k0;
    if (kt == 32) {
#pragma unroll
      for (int q = 0; q < 32; ++q) acc[q] += g * wr[q];
    } else {
      for (int q = 0; q < kt; ++q) acc[q] += g * wr[q];
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const float v = wave_sum(acc[q]);
    if (lane == 0) s_red[wv][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < kt) {
    const float v = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    const long o = (long)b * K + k0 + threadIdx.x;
    dx[o] = v * act_grad_f(x[o], act);
  }
}
// dbias[n] += sum_b dy[b][n]: 64 columns x 4 row groups per block, 256 rows per block, one atomic per column per block
__global__ __launch_bounds__(256) void linear_bwd_dbias_kernel(const float* __restrict__ dy, float* __restrict__ dbias, int Bn, int N) {
  __shared__ float s_red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * 256, r1 = min(Bn, r0 + 256);
  float acc = 0.f;
  if (n < N)
    for (int b = r0 + ty; b < r1; b += 4) acc += dy[(long)b * N + n];
  s_red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && n < N) atomicAdd(&dbias[n], s_red[0][tx] + s_red[1][tx] + s_red[2][tx] + s_red[3][tx]);
}

__global__ void gamma_embedding_kernel(const float* __restrict__ gammas, float* __restrict__ emb, int Bn, int dim,
                                       float neg_log_period) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Bn * dim) return;
  const int b = idx / dim, j = idx % dim;
  const int half = dim / 2;
  if (j >= 2 * half) { emb[idx] = 0.f; return; }
  const int k = j < half ? j : j - half;
  const float freq = expf(neg_log_period * (float)k / (float)half);
  const float arg = gammas[b] * freq;
  emb[idx] = j < half ? cosf(arg) : sinf(arg);
}

// ---- DDPM glue ---------------------------------------------------------------------------------
template <typename T>
__global__ void ddpm_prepare_kernel(const float* __restrict__ y0, const float* __restrict__ ycond,
                                    const float* __restrict__ noise, const int64_t* __restrict__ mask,
                                    const float* __restrict__ gammas, T* __restrict__ xin, int B, int C, int HW, int Cpad) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = i / HW;
    const long p = i % HW;
    const float g = gammas[b];
    const float sg = sqrtf(g), s1 = sqrtf(1.0f - g);
    float m = 1.0f;
    if (mask) {
      const int64_t mv = mask[i];
      m = mv < 0 ? 0.f : (mv > 1 ? 1.f : (float)mv);
    }
    T* o = xin + i * Cpad;
    for (int c = 0; c < Cpad; ++c) {
      float v = 0.f;
      if (c < C) {
        v = ycond[((long)b * C + c) * HW + p];
      } else if (c < 2 * C) {
        const long q = ((long)b * C + (c - C)) * HW + p;
        const float yv = y0[q];
        float yn = sg * yv + s1 * noise[q];
        if (mask) yn = yn * m + (1.0f - m) * yv;
        v = yn;
      }
      o[c] = from_f32<T>(v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ddpm_mse_loss_kernel(const float* __restrict__ noise, const T* __restrict__ nh,
                                                            const int64_t* __restrict__ mask, const float* __restrict__ w,
                                                            float* __restrict__ loss, T* __restrict__ dnh, int B, int C,
                                                            int HW, int Cpad, float lambda, float grad_scale) {
  __shared__ float s_part[4];
  const long total = (long)B * HW;
  const float invN = 1.0f / ((float)total * (float)C);
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = i / HW;
    const long p = i % HW;
    float wm = w ? w[b] : 1.0f;
    if (mask) {
      const int64_t mv = mask[i];
      wm *= mv < 0 ? 0.f : (mv > 1 ? 1.f : (float)mv);
    }
    for (int c = 0; c < Cpad; ++c) {
      float gout = 0.f;
      if (c < C) {
        const float n = noise[((long)b * C + c) * HW + p];
        const float h = to_f32(nh[i * Cpad + c]);
        const float d = wm * n - wm * h;
        acc += d * d;
        gout = -2.0f * d * wm * invN * lambda * grad_scale;
      }
      if (dnh) dnh[i * Cpad + c] = from_f32<T>(gout);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * invN * lambda);
}

// ---- L1 / multiscale diffusion losses (models/modules/loss.py:397-467, palette_model.py:231-256,597-618) ---------------------
// d = w m (noise - noise_hat).  F.interpolate(size = S / f, mode="bilinear", align_corners=False) with an integer factor f >= 2
// samples at (i + 0.5) f - 0.5: the mean of the CENTRAL 2x2 pixels of every f x f cell (rows f i + f/2 - 1, f i + f/2).
// Level l has factor f = 1 << l.  Pass 1 writes the down-sampled d of the levels l >= 1 into `ws` and their loss sums;
// pass 2 handles level 0 and scatters every level's derivative back to the full-resolution gradient.
__device__ __forceinline__ float ms_g(float x, int l1) { return l1 ? fabsf(x) : x * x; }
__device__ __forceinline__ float ms_dg(float x, int l1) { return l1 ? (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f)) : 2.0f * x; }
__device__ __forceinline__ float ms_wm(const int64_t* mask, const float* w, int b, long pix) {
  float wm = w ? w[b] : 1.0f;
  if (mask) {
    const int64_t mv = mask[pix];
    wm *= mv < 0 ? 0.f : (mv > 1 ? 1.f : (float)mv);
  }
  return wm;
}
__device__ __forceinline__ long ms_level_offset(int l, int B, int C, int H, int W) {   // elements of levels 1 .. l-1
  long off = 0;
  for (int k = 1; k < l; ++k) off += (long)B * C * (H >> k) * (W >> k);
  return off;
}

template <typename T>
__global__ __launch_bounds__(256) void ms_loss_down_kernel(const float* __restrict__ noise, const T* __restrict__ nh,
                                                           const int64_t* __restrict__ mask, const float* __restrict__ w,
                                                           float* __restrict__ ws, float* __restrict__ losses, int B, int C, int H, int W,
                                                           int Cpad, int nlevels, int l1, float lambda) {
  __shared__ float s_part[4];
  const int l = blockIdx.y + 1;
  const int f = 1 << l, Hl = H >> l, Wl = W >> l;
  const long total = (long)B * C * Hl * Wl;
  float* out = ws + ms_level_offset(l, B, C, H, W);
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int J = i % Wl;
    long r = i / Wl;
    const int I = r % Hl;
    r /= Hl;
    const int c = r % C, b = r / C;
    const int y0 = f * I + f / 2 - 1, x0 = f * J + f / 2 - 1;
    float v = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const long pix = ((long)b * H + y0 + dy) * W + x0 + dx;
        const float wm = ms_wm(mask, w, b, pix);
        v += wm * noise[((long)b * C + c) * H * W + (long)(y0 + dy) * W + x0 + dx] - wm * to_f32(nh[pix * Cpad + c]);
      }
    v *= 0.25f;
    out[i] = v;
    acc += ms_g(v, l1);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  // weight of the level: min_res / (2 res) with min_res = 32, res = W / f   (loss.py:440-447,463-465)
  if (threadIdx.x == 0) atomicAdd(losses + l, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)total * (16.0f * f / (float)W) * lambda);
}

template <typename T>
__global__ __launch_bounds__(256) void ms_loss_grad_kernel(const float* __restrict__ noise, const T* __restrict__ nh,
                                                           const int64_t* __restrict__ mask, const float* __restrict__ w,
                                                           const float* __restrict__ ws, float* __restrict__ losses, T* __restrict__ dnh,
                                                           int B, int C, int H, int W, int Cpad, int nlevels, int l1, int multiscale,
                                                           float lambda, float grad_scale) {
  __shared__ float s_part[4];
  const long HW = (long)H * W, total = (long)B * HW;
  const float c0 = (multiscale ? 16.0f / (float)W : 1.0f) * lambda / ((float)total * (float)C);
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = i / HW;
    const long p = i % HW;
    const int y = p / W, x = p % W;
    const float wm = ms_wm(mask, w, b, i);
    for (int c = 0; c < Cpad; ++c) {
      float gout = 0.f;
      if (c < C) {
        const float d = wm * noise[((long)b * C + c) * HW + p] - wm * to_f32(nh[i * Cpad + c]);
        acc += ms_g(d, l1);
        float g = c0 * ms_dg(d, l1);
        long off = 0;
        for (int l = 1; l < nlevels; ++l) {
          const int f = 1 << l, Hl = H >> l, Wl = W >> l;
          const int ry = y & (f - 1), rx = x & (f - 1);
          if ((ry == f / 2 - 1 || ry == f / 2) && (rx == f / 2 - 1 || rx == f / 2)) {
            const float v = ws[off + (((long)b * C + c) * Hl + (y >> l)) * Wl + (x >> l)];
            g += 0.25f * ms_dg(v, l1) * (16.0f * f / (float)W) * lambda / ((float)B * C * Hl * Wl);
          }
          off += (long)B * C * Hl * Wl;
        }
        gout = -wm * g * grad_scale;
      }
      if (dnh) dnh[i * Cpad + c] = from_f32<T>(gout);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(losses, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * c0);
}

// ---- CUT / ResNet-generator glue (resnet_generator.py, discriminators.py) ------------------------------------
// standalone activations on 16-byte vectors: y = act(x); backward from the OUTPUT y (tanh: 1 - y^2; (leaky) relu: the
// sign of y is the sign of x)
template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n8, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(x + i * 8), f);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      f[q] = act == JG_ACT_TANH ? tanhf(f[q]) : act == JG_ACT_RELU ? fmaxf(f[q], 0.f) : act == JG_ACT_LRELU ? (f[q] > 0.f ? f[q] : 0.2f * f[q])
                                : act == JG_ACT_SILU ? silu_f(f[q]) : f[q];
    *reinterpret_cast<uint4*>(y + i * 8) = pack8<T>(f);
  }
}
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, long n8, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(y + i * 8), f);
    unpack8<T>(*reinterpret_cast<const uint4*>(dy + i * 8), g);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      g[q] *= act == JG_ACT_TANH ? (1.f - f[q] * f[q]) : act == JG_ACT_RELU ? (f[q] > 0.f ? 1.f : 0.f) : act == JG_ACT_LRELU ? (f[q] > 0.f ? 1.f : 0.2f) : 1.f;
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8<T>(g);
  }
}

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// interior crop of an NHWC tensor (y = x[:, top:top+Ho, left:left+Wo]) and its adjoint (zero outside the window): the mobile
// (depth-wise separable) residual blocks run reflect-pad -> zero-padded depth-wise conv -> crop
template <typename T, bool ADJ>
__global__ void crop_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int C, int top, int left, int Ho, int Wo) {
  const int noct = C / 8;
  const int Hd = ADJ ? H : Ho, Wd = ADJ ? W : Wo;
  const long total = (long)B * Hd * Wd * noct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % noct;
    long t = i / noct;
    const int w = t % Wd; t /= Wd;
    const int h = t % Hd;
    const int b = t / Hd;
    if (ADJ) {
      const int oh = h - top, ow = w - left;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (oh >= 0 && oh < Ho && ow >= 0 && ow < Wo) v = *reinterpret_cast<const uint4*>(src + ((((long)b * Ho + oh) * Wo + ow) * noct + co) * 8);
      *reinterpret_cast<uint4*>(dst + i * 8) = v;
    } else {
      *reinterpret_cast<uint4*>(dst + i * 8) = *reinterpret_cast<const uint4*>(src + ((((long)b * H + h + top) * W + w + left) * noct + co) * 8);
    }
  }
}
// nn.ReflectionPad2d(pad), NHWC: y[b, i, j] = x[b, refl(i - pad), refl(j - pad)]
template <typename T>
__global__ void reflect_pad_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int pad) {
  const int Ho = H + 2 * pad, Wo = W + 2 * pad, noct = C / 8;
  const long total = (long)B * Ho * Wo * noct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % noct;
    long t = i / noct;
    const int ow = t % Wo; t /= Wo;
    const int oh = t % Ho;
    const int b = t / Ho;
    const int ih = reflect_idx(oh - pad, H), iw = reflect_idx(ow - pad, W);
    *reinterpret_cast<uint4*>(y + i * 8) = *reinterpret_cast<const uint4*>(x + (((long)b * H + ih) * W + iw) * C + co * 8);
  }
}
// its adjoint as a gather: dx[b, i, j] = sum of dy over the (<= 2 x 2) padded positions that mirror onto (i, j)
template <typename T>
__global__ void reflect_pad_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C, int pad) {
  const int Ho = H + 2 * pad, Wo = W + 2 * pad, noct = C / 8;
  const long total = (long)B * H * W * noct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % noct;
    long t = i / noct;
    const int iw = t % W; t /= W;
    const int ih = t % H;
    const int b = t / H;
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = ih + pad;
    if (ih >= 1 && ih <= pad) hs[nh++] = pad - ih;
    if (ih <= H - 2 && ih >= H - 1 - pad) hs[nh++] = 2 * (H - 1) - ih + pad;
    ws[nw++] = iw + pad;
    if (iw >= 1 && iw <= pad) ws[nw++] = pad - iw;
    if (iw <= W - 2 && iw >= W - 1 - pad) ws[nw++] = 2 * (W - 1) - iw + pad;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < nh; ++a)
      for (int c = 0; c < nw; ++c) {
        float f[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(dy + (((long)b * Ho + hs[a]) * Wo + ws[c]) * C + co * 8), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += f[q];
      }
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8<T>(acc);
  }
}

// zero insertion y[b, s*i, s*j] = x[b, i, j] into a [B, Ho, Wo] buffer (0 elsewhere): the stride-s transposed convolution /
// the input gradient of a stride-s convolution is a stride-1 convolution of this buffer with the flipped weights.
// `gather` = the adjoint: y[b, i, j] = x[b, s*i, s*j].
template <typename T>
__global__ void dilate_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo, int s) {
  const int noct = C / 8;
  const long total = (long)B * Ho * Wo * noct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % noct;
    long t = i / noct;
    const int ow = t % Wo; t /= Wo;
    const int oh = t % Ho;
    const int b = t / Ho;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (oh % s == 0 && ow % s == 0 && oh / s < H && ow / s < W)
      v = *reinterpret_cast<const uint4*>(x + (((long)b * H + oh / s) * W + ow / s) * C + co * 8);
    *reinterpret_cast<uint4*>(y + i * 8) = v;
  }
}
template <typename T>
__global__ void subsample_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo, int s) {
  const int noct = C / 8;
  const long total = (long)B * Ho * Wo * noct;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = i % noct;
    long t = i / noct;
    const int ow = t % Wo; t /= Wo;
    const int oh = t % Ho;
    const int b = t / Ho;
    *reinterpret_cast<uint4*>(y + i * 8) = *reinterpret_cast<const uint4*>(x + (((long)b * H + oh * s) * W + ow * s) * C + co * 8);
  }
}

// out[c] += scale * sum over P pixels of x[p][c]  (bias gradient of a transposed convolution)
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out, long P,
                                                          int C, float scale) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int lane_p = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C)
    for (long p = blockIdx.x * 4L + lane_p; p < P; p += (long)gridDim.x * 4) acc += to_f32(x[p * ldx + c]);
  __shared__ float sacc[4][64];
  sacc[lane_p][threadIdx.x & 63] = acc;
  __syncthreads();
  if (lane_p == 0 && c < C) atomicAdd(out + c, scale * (sacc[0][threadIdx.x] + sacc[1][threadIdx.x] + sacc[2][threadIdx.x] + sacc[3][threadIdx.x]));
}

// the same with 16-byte loads (round 6): a thread owns 8 channels and strides over pixels, four loads in flight; the block's pixel lanes are summed
// through LDS and ONE atomic per channel leaves the block.  (The form above reads one 2-byte element per lane and pixel: 417 us for the 268 MB
// of a ConvTranspose bias gradient at 256 x 256 x 64 channels x 32 images, 15 x its bytes' time.)  C a multiple of 8, <= 2048; ldx a multiple of 8.
template <typename T>
__global__ __launch_bounds__(256) void channel_sum8_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out, long P, int C, float scale) {
  extern __shared__ float s_cs[];          // [pixel lanes][C]
  const int noct = C >> 3;
  const int pl = 256 / noct > 0 ? 256 / noct : 1;
  const int co = threadIdx.x % noct, lp = threadIdx.x / noct;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = 0.f;
  if (lp < pl) {
    const long stride = (long)gridDim.x * pl;
    long p = blockIdx.x * (long)pl + lp;
    for (; p + 3 * stride < P; p += 4 * stride) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(x + (p + u * stride) * ldx + co * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8<T>(v[u], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += f[q];
      }
    }
    for (; p < P; p += stride) {
      float f[8];
      unpack8<T>(*reinterpret_cast<const uint4*>(x + p * ldx + co * 8), f);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += f[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) s_cs[lp * C + co * 8 + q] = acc[q];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = 0.f;
    for (int l = 0; l < pl; ++l) v += s_cs[l * C + c];
    atomicAdd(out + c, scale * v);
  }
}

// ---- PatchSampleF / GAN-loss glue (cut_networks.py:6-73, loss.py:59-85) ------------------------------------------
// dst[b*P + p][c] = src[b, ids[p], c] as fp32 (the SAME patch ids for every image of the batch, cut_networks.py:43-57);
// scatter = its adjoint (ids come from randperm: unique, so plain stores into a zeroed gradient).
// Grouped form (G id sets, round 5): image b reads ids[((b / per) % G) * P + p] -- consecutive runs of `per` images cycle through the G sets,
// so ONE launch serves the concatenated batch [translated | identity | source | target] of the two contrastive terms (G = 2, per = B).
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, long ld, const int64_t* __restrict__ ids, float* __restrict__ dst,
                                   int B, long HW, int C, int P, int G, int per) {
  const long total = (long)B * P * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % C;
    const long r = i / C;
    const int p = r % P, b = r / P;
    dst[i] = to_f32(src[((long)b * HW + ids[((b / per) % G) * P + p]) * ld + c]);
  }
}
template <typename T>
__global__ void scatter_rows_kernel(T* __restrict__ dsrc, long ld, const int64_t* __restrict__ ids, const float* __restrict__ ddst,
                                    int B, long HW, int C, int P, int G, int per) {
  const long total = (long)B * P * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % C;
    const long r = i / C;
    const int p = r % P, b = r / P;
    dsrc[((long)b * HW + ids[((b / per) % G) * P + p]) * ld + c] = from_f32<T>(ddst[i]);
  }
}

// torch.nn.functional.normalize(x, eps): y = x / max(||x||_2, eps) per row; one wave per row
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ nrm,
                                                         long R, int D, float eps) {
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = x[row * D + d]; acc += v * v; }
  acc = wave_sum(acc);
  const float n = fmaxf(sqrtf(acc), eps);
  if (lane == 0) nrm[row] = n;
  for (int d = lane; d < D; d += 64) y[row * D + d] = x[row * D + d] / n;
}
// dx = (dy - y (y . dy)) / n   (rows with ||x|| <= eps: dx = dy / eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ nrm,
                                                         const float* __restrict__ dy, float* __restrict__ dx, long R, int D,
                                                         float eps) {
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) acc += y[row * D + d] * dy[row * D + d];
  acc = wave_sum(acc);
  const float n = nrm[row];
  const bool clamped = n <= eps;
  for (int d = lane; d < D; d += 64) dx[row * D + d] = (dy[row * D + d] - (clamped ? 0.f : y[row * D + d] * acc)) / n;
}

// GANLoss (loss.py:59-76) on the FIRST channel of a [*, Cpad] logit map, loss += scale * mean(f(pred)), dpred = grad_scale * scale * f'(pred) / N in
// channel 0 and zero in the padding channels.  MODE 0 "lsgan": f = (x - target)^2 (nn.MSELoss);  1 "vanilla": f = BCE-with-logits against
// the label `target` = max(x, 0) - x target + log(1 + exp(-|x|)) (nn.BCEWithLogitsLoss);  2 "wgangp": f = -x for real (target >= 0.5), +x for fake
// (the reference never adds its gradient penalty: cal_gradient_penalty has no caller).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void gan_loss_kernel(const T* __restrict__ pred, float target, float* __restrict__ loss,
                                                       T* __restrict__ dpred, long Npix, int Cpad, float scale, float grad_scale) {
  __shared__ float s_part[4];
  float acc = 0.f;
  const float invN = 1.0f / (float)Npix;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < Npix; i += (long)gridDim.x * blockDim.x) {
    const float x = to_f32(pred[i * Cpad]);
    float f, df;
    if (MODE == 0) {
      const float d = x - target;
      f = d * d;
      df = 2.0f * d;
    } else if (MODE == 1) {
      const float e = expf(-fabsf(x));
      f = fmaxf(x, 0.f) - x * target + log1pf(e);
      df = (x >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e)) - target;      // sigmoid(x) - target
    } else {
      const float sgn = target >= 0.5f ? -1.0f : 1.0f;
      f = sgn * x;
      df = sgn;
    }
    acc += f;
    if (dpred) {
      dpred[i * Cpad] = from_f32<T>(grad_scale * scale * df * invN);
      for (int c = 1; c < Cpad; ++c) dpred[i * Cpad + c] = from_f32<T>(0.f);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * invN * scale);
}

// ---- DDPM ancestral sampling step (diffusion_generator.py:187-284, diffusion_utils.py:122-137) -------------
// One reverse step after the UNet: y0_hat = clamp(sr*y_t - srm1*noise_hat, -1, 1); mean = c1*y0_hat + c2*y_t;
// y' = mean + z * exp(0.5*logvar); with a mask y' = y_0*(1-m) + m*y', m = clamp(mask,0,1).  Writes y' (fp32 NCHW,
// in place over y_t) and the next UNet input [y_cond | y' | 0-pad] (16-bit NHWC).  coef[b] = {sr, srm1, c1, c2, sigma}.
template <typename T>
__global__ void ddpm_p_sample_kernel(float* __restrict__ y_t, const float* __restrict__ y_cond, const T* __restrict__ nh,
                                     const float* __restrict__ z, const float* __restrict__ y_0,
                                     const int64_t* __restrict__ mask, const float* __restrict__ coef, T* __restrict__ xin,
                                     int B, int C, int HW, int Cpad_in, int Cpad_out, int clip) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = i / HW;
    const long p = i % HW;
    const float sr = coef[b * 5], srm1 = coef[b * 5 + 1], c1 = coef[b * 5 + 2], c2 = coef[b * 5 + 3], sg = coef[b * 5 + 4];
    float m = 1.0f;
    if (mask) {
      const int64_t mv = mask[i];
      m = mv < 0 ? 0.f : (mv > 1 ? 1.f : (float)mv);
    }
    T* o = xin + i * Cpad_out;
    for (int c = 0; c < Cpad_out; ++c) {
      float v = 0.f;
      if (c < C) {
        v = y_cond[((long)b * C + c) * HW + p];
      } else if (c < 2 * C) {
        const long q = ((long)b * C + (c - C)) * HW + p;
        const float yt = y_t[q];
        float y0h = sr * yt - srm1 * to_f32(nh[i * Cpad_in + (c - C)]);
        if (clip & 1) y0h = fminf(fmaxf(y0h, -1.0f), 1.0f);
        float yn = c1 * y0h + c2 * yt + (z ? z[q] * sg : 0.f);
        if (clip & 2) yn = fminf(fmaxf(yn, -1.0f), 1.0f);   // DDIM: the reference also clamps the mean (:452-453)
        if (mask) yn = y_0[q] * (1.0f - m) + m * yn;
        y_t[q] = yn;
        v = yn;
      }
      o[c] = from_f32<T>(v);
    }
  }
}

// ---- consistency-model glue (cm_generator.py:367-502, cm_model.py:27-43,353-375) -------------------------
// noisy = x + sigma[b] * noise; with a mask: noisy * clamp(mask,0,1) + (1 - clamp(mask,0,1)) * x.  Written twice:
// fp32 NCHW (the c_skip * x term / visuals) and 16-bit NHWC with `cond` (optional, Ccond channels) in front
// = torch.cat([x_cond, x], 1), zero padded to Cpad: the UNet input.
template <typename T>
__global__ void cm_noisy_kernel(const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ sigma,
                                const int64_t* __restrict__ mask, const float* __restrict__ cond, float* __restrict__ out_nchw,
                                T* __restrict__ out_nhwc, int B, int C, int Ccond, int HW, int Cpad) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = i / HW;
    const long p = i % HW;
    const float sg = sigma[b];
    float m = 1.0f;
    if (mask) {
      const int64_t mv = mask[i];
      m = mv < 0 ? 0.f : (mv > 1 ? 1.f : (float)mv);
    }
    T* o = out_nhwc + i * Cpad;
    for (int c = 0; c < Cpad; ++c) {
      float v = 0.f;
      if (c < Ccond) {
        v = cond[((long)b * Ccond + c) * HW + p];
      } else if (c < Ccond + C) {
        const long q = ((long)b * C + (c - Ccond)) * HW + p;
        const float xv = x[q];
        float nv = xv + sg * noise[q];
        if (mask) nv = nv * m + (1.0f - m) * xv;
        out_nchw[q] = nv;
        v = nv;
      }
      o[c] = from_f32<T>(v);
    }
  }
}

// out[b,c,p] = c_skip[b] * noisy[b,c,p] + c_out[b] * F[b,p,c]     (cm_forward, cm_generator.py:383-385)
template <typename T>
__global__ void cm_combine_kernel(const float* __restrict__ noisy, const T* __restrict__ F, const float* __restrict__ cskip,
                                  const float* __restrict__ cout, float* __restrict__ out, int B, int C, int HW, int Cpad) {
  const long total = (long)B * C * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % HW;
    const long r = i / HW;
    const int c = r % C;
    const int b = r / C;
    out[i] = cskip[b] * noisy[i] + cout[b] * to_f32(F[((long)b * HW + p) * Cpad + c]);
  }
}

// Consistency loss with its gradient in one pass (compute_cm_loss, cm_model.py:353-375 + pseudo_huber_loss :27-43):
//   pred = cs_n*noisy_n + co_n*F_n (student),  target = cs_c*noisy_c + co_c*F_c (no-grad teacher),
//   loss = lambda * mean( w[b] * (sqrt((m*pred - m*target)^2 + c^2) - c) ),  m = the label mask AS IS (not clamped)
//   dF_n = grad_scale * lambda * w[b]/N * d/sqrt(d^2+c^2) * m * co_n
template <typename T>
__global__ __launch_bounds__(256) void cm_loss_kernel(const T* __restrict__ Fn, const T* __restrict__ Fc,
                                                      const float* __restrict__ noisy_n, const float* __restrict__ noisy_c,
                                                      const float* __restrict__ cs_n, const float* __restrict__ co_n,
                                                      const float* __restrict__ cs_c, const float* __restrict__ co_c,
                                                      const int64_t* __restrict__ mask, const float* __restrict__ w,
                                                      float* __restrict__ loss, T* __restrict__ dFn, int B, int C, int HW,
                                                      int Cpad, float chub, float lambda, float grad_scale) {
  __shared__ float s_part[4];
  const long total = (long)B * HW;
  const float invN = 1.0f / ((float)total * (float)C);
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = i / HW;
    const long p = i % HW;
    const float m = mask ? (float)mask[i] : 1.0f;
    const float wb = w[b];
    for (int c = 0; c < Cpad; ++c) {
      float gout = 0.f;
      if (c < C) {
        const long q = ((long)b * C + c) * HW + p;
        const float pred = cs_n[b] * noisy_n[q] + co_n[b] * to_f32(Fn[i * Cpad + c]);
        const float targ = cs_c[b] * noisy_c[q] + co_c[b] * to_f32(Fc[i * Cpad + c]);
        const float d = m * pred - m * targ;
        const float r = sqrtf(d * d + chub * chub);
        acc += wb * (r - chub);
        gout = grad_scale * lambda * invN * wb * (d / r) * m * co_n[b];
      }
      if (dFn) dFn[i * Cpad + c] = from_f32<T>(gout);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * invN * lambda);
}

// NoiseLevelEmbedding (cm_generator.py:276-280): h = sigma * W * 2 * pi; [sin(h) | cos(h)]
__global__ void noise_level_embedding_kernel(const float* __restrict__ sigma, const float* __restrict__ W, float* __restrict__ emb,
                                             int Bn, int half) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Bn * 2 * half) return;
  const int b = idx / (2 * half), j = idx % (2 * half);
  const int k = j < half ? j : j - half;
  const float h = ((sigma[b] * W[k]) * 2.0f) * 3.14159265358979323846f;
  emb[idx] = j < half ? sinf(h) : cosf(h);
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C, int HW, int Cpad) {
  const long total = (long)B * C * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % HW;
    const long r = i / HW;
    const int c = r % C;
    const int b = r / C;
    y[i] = to_f32(x[((long)b * HW + p) * Cpad + c]);
  }
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int HW, int Cpad) {
  const long total = (long)B * HW * Cpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % Cpad;
    const long r = i / Cpad;
    const long p = r % HW;
    const int b = r / HW;
    y[i] = from_f32<T>(c < C ? x[((long)b * C + c) * HW + p] : 0.f);
  }
}

// loss reductions (one atomicAdd of a workgroup partial per workgroup): JG_DETERMINISTIC 1 -> ONE workgroup walks everything in its
// grid-stride loop, the wave partials are summed in a fixed order: a reproducible loss
inline int grid_for(long total, int block, int cap);
inline int loss_grid(long total, int block, int cap) { return jg_tune(JG_TUNE_DETERMINISTIC) != 0 ? 1 : grid_for(total, block, cap); }
inline int grid_for(long total, int block = 256, int cap = 256 * 16) {
  long g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace

extern "C" int jg_pool2x2_ld(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C,
                             float scale, jg_stream_t s) {
  if (!x || !y || C % 8 || H % 2 || W % 2 || B < 1 || ldx < C || ldy < C || ldx % 8 || ldy % 8) return JG_ERR_BAD_ARG;
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pool2x2_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)x, (long)ldx, (T*)y, (long)ldy, B, H, W, C, scale););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_pool2x2(int dtype, const void* x, void* y, int B, int H, int W, int C, float scale, jg_stream_t s) {
  return jg_pool2x2_ld(dtype, x, C, y, C, B, H, W, C, scale, s);
}
extern "C" int jg_upsample2x_ld(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C,
                                float scale, jg_stream_t s) {
  if (!x || !y || C % 8 || B < 1 || ldx < C || ldy < C || ldx % 8 || ldy % 8) return JG_ERR_BAD_ARG;
  const long total = (long)B * H * 2 * W * 2 * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((upsample2x_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)x, (long)ldx, (T*)y, (long)ldy, B, H, W, C, scale););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_upsample2x(int dtype, const void* x, void* y, int B, int H, int W, int C, float scale, jg_stream_t s) {
  return jg_upsample2x_ld(dtype, x, C, y, C, B, H, W, C, scale, s);
}
extern "C" int jg_copy_channels(int dtype, const void* src, int64_t ldsrc, int64_t soff, void* dst, int64_t lddst,
                                int64_t doff, int64_t P, int n, jg_stream_t s) {
  if (!src || !dst || n % 8 || ldsrc % 8 || lddst % 8 || soff % 8 || doff % 8 || P < 1) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((copy_channels_kernel<T>), dim3(grid_for(P * (n / 8))), dim3(256), 0,
                                              (hipStream_t)s, (const T*)src, (long)ldsrc, (long)soff, (T*)dst, (long)lddst,
                                              (long)doff, (long)P, n););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_axpby(int dtype, const void* a, float alpha, const float* alpha_dev, const void* b, float beta, void* y,
                        int64_t n, jg_stream_t s) {
  if (!a || !y || n % 8) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((axpby_kernel<T>), dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)a, alpha, alpha_dev, (const T*)b, beta, (T*)y, (long)(n / 8)););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_transpose_heads(int dtype, const void* src, int64_t ldsrc, int64_t coff, int64_t hstride, void* dst,
                                  int B, int Tn, int nh, int ch, jg_stream_t s) {
  if (!src || !dst || ch % 8 || ldsrc % 8 || coff % 8 || hstride % 8) return JG_ERR_BAD_ARG;
  const long total = (long)B * nh * (ch / 8) * Tn;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose_heads_kernel<T>), dim3(grid_for(total)), dim3(256), 0,
                                              (hipStream_t)s, (const T*)src, (long)ldsrc, (long)coff, (long)hstride,
                                              (T*)dst, B, Tn, nh, ch););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_softmax_fwd(int dtype, const float* S, void* P, int64_t rows, int Tn, jg_stream_t s) {
  if (!S || !P || Tn % 4 || rows < 1) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((softmax_fwd_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                              (hipStream_t)s, S, (T*)P, (long)rows, Tn););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_softmax_bwd(int dtype, const void* P, const float* dP, void* dS, int64_t rows, int Tn, float alpha,
                              jg_stream_t s) {
  if (!P || !dP || !dS || Tn % 4 || rows < 1) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((softmax_bwd_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                              (hipStream_t)s, (const T*)P, dP, (T*)dS, (long)rows, Tn, alpha););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
// nn.Linear on the fp32 GEMM of nce.hip: y = act(x) W^T + b; dx = act'(x) .* (dy W); dW += dy^T act(x); db += sum_b dy
extern "C" int jg_linear_fwd(const float* x, const float* W, const float* bias, float* y, int Bn, int K, int N, int act,
                             jg_stream_t s) {
  if (!x || !W || !y || Bn < 1 || K < 1 || N < 1) return JG_ERR_BAD_ARG;
  return jg_sgemm(x, W, y, bias, nullptr, Bn, N, K, K, 1, K, 1, N, 1, 1, 0, 0, 0, 1.0f, 0.0f, act, JG_ACT_NONE, JG_ACT_NONE, s);
}
extern "C" int jg_linear_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* dbias, int Bn,
                             int K, int N, int act, jg_stream_t s) {
  if (!x || !W || !dy || Bn < 1 || K < 1 || N < 1) return JG_ERR_BAD_ARG;
  int rc = JG_OK;
  if (dx && Bn <= 64 && N >= 2048)     // 4 output tiles and a reduction of thousands: one block per (row, 32 columns) instead
    hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(Bn, (K + 31) / 32), dim3(256), 0, (hipStream_t)s, x, W, dy, dx, Bn, K, N, act);
  else if (dx) rc = jg_sgemm(dy, W, dx, nullptr, act == JG_ACT_NONE ? nullptr : x, Bn, K, N, N, 1, 1, K, K, 1, 1, 0, 0, 0, 1.0f, 0.0f,
                             JG_ACT_NONE, JG_ACT_NONE, act, s);
  if (rc != JG_OK) return rc;
  if (dW) rc = jg_sgemm(dy, x, dW, nullptr, nullptr, N, K, Bn, 1, N, 1, K, K, 1, 1, 0, 0, 0, 1.0f, 1.0f, JG_ACT_NONE, act, JG_ACT_NONE, s);
  if (rc != JG_OK) return rc;
  if (dbias)
    hipLaunchKernelGGL(linear_bwd_dbias_kernel, dim3((N + 63) / 64, (Bn + 255) / 256), dim3(256), 0, (hipStream_t)s, dy, dbias, Bn, N);
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_gamma_embedding(const float* gammas, float* emb, int Bn, int dim, float max_period, jg_stream_t s) {
  if (!gammas || !emb || Bn < 1 || dim < 2) return JG_ERR_BAD_ARG;
  const float nlp = -(float)log((double)max_period);
  hipLaunchKernelGGL(gamma_embedding_kernel, dim3((Bn * dim + 255) / 256), dim3(256), 0, (hipStream_t)s, gammas, emb, Bn, dim, nlp);
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_ddpm_prepare(int dtype, const float* y0, const float* ycond, const float* noise, const int64_t* mask,
                               const float* gammas, void* xin, int B, int C, int H, int W, int Cpad, jg_stream_t s) {
  if (!y0 || !ycond || !noise || !gammas || !xin || Cpad < 2 * C || Cpad % 8) return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ddpm_prepare_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                              y0, ycond, noise, mask, gammas, (T*)xin, B, C, H * W, Cpad););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_ddpm_mse_loss(int dtype, const float* noise, const void* noise_hat, const int64_t* mask, const float* w,
                                float* loss, void* dnh, int B, int C, int H, int W, int Cpad, float lambda,
                                float grad_scale, jg_stream_t s) {
  if (!noise || !noise_hat || !loss || Cpad < C || Cpad % 8) return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ddpm_mse_loss_kernel<T>), dim3(loss_grid(total, 256, 1024)), dim3(256), 0,
                                              (hipStream_t)s, noise, (const T*)noise_hat, mask, w, loss, (T*)dnh, B, C,
                                              H * W, Cpad, lambda, grad_scale););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_nhwc_to_nchw_f32(int dtype, const void* x, float* y, int B, int C, int H, int W, int Cpad, jg_stream_t s) {
  if (!x || !y || Cpad < C) return JG_ERR_BAD_ARG;
  const long total = (long)B * C * H * W;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)x, y, B, C, H * W, Cpad););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_nchw_f32_to_nhwc(int dtype, const float* x, void* y, int B, int C, int H, int W, int Cpad, jg_stream_t s) {
  if (!x || !y || Cpad < C) return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W * Cpad;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                              x, (T*)y, B, C, H * W, Cpad););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_cm_noisy(int dtype, const float* x, const float* noise, const float* sigma, const int64_t* mask,
                           const float* cond, float* out_nchw, void* out_nhwc, int B, int C, int Ccond, int H, int W, int Cpad,
                           jg_stream_t s) {
  if (!x || !noise || !sigma || !out_nchw || !out_nhwc || Cpad < C + Ccond || Cpad % 8 || (Ccond > 0 && !cond)) return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cm_noisy_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, x, noise,
                                              sigma, mask, cond, out_nchw, (T*)out_nhwc, B, C, Ccond, H * W, Cpad););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_cm_combine(int dtype, const float* noisy, const void* F, const float* cskip, const float* cout, float* out,
                             int B, int C, int H, int W, int Cpad, jg_stream_t s) {
  if (!noisy || !F || !cskip || !cout || !out || Cpad < C || Cpad % 8) return JG_ERR_BAD_ARG;
  const long total = (long)B * C * H * W;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cm_combine_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, noisy,
                                              (const T*)F, cskip, cout, out, B, C, H * W, Cpad););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_cm_loss(int dtype, const void* Fn, const void* Fc, const float* noisy_n, const float* noisy_c,
                          const float* cs_n, const float* co_n, const float* cs_c, const float* co_c, const int64_t* mask,
                          const float* w, float* loss, void* dFn, int B, int C, int H, int W, int Cpad, float c_huber,
                          float lambda, float grad_scale, jg_stream_t s) {
  if (!Fn || !Fc || !noisy_n || !noisy_c || !cs_n || !co_n || !cs_c || !co_c || !w || !loss || Cpad < C || Cpad % 8)
    return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cm_loss_kernel<T>), dim3(loss_grid(total, 256, 1024)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)Fn, (const T*)Fc, noisy_n, noisy_c, cs_n, co_n, cs_c, co_c, mask, w, loss,
                                              (T*)dFn, B, C, H * W, Cpad, c_huber, lambda, grad_scale););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
// dW[k] += sum_b 2 pi sigma_b (demb[b][k] cos(h) - demb[b][half + k] sin(h)),  h = sigma_b W_k 2 pi
__global__ void noise_level_embedding_bwd_kernel(const float* __restrict__ sigma, const float* __restrict__ W,
                                                 const float* __restrict__ demb, float* __restrict__ dW, int Bn, int half) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= half) return;
  float acc = 0.f;
  for (int b = 0; b < Bn; ++b) {
    const float two_pi_s = (sigma[b] * 2.0f) * 3.14159265358979323846f;
    const float h = ((sigma[b] * W[k]) * 2.0f) * 3.14159265358979323846f;
    acc += two_pi_s * (demb[(long)b * 2 * half + k] * cosf(h) - demb[(long)b * 2 * half + half + k] * sinf(h));
  }
  atomicAdd(dW + k, acc);
}

extern "C" int jg_noise_level_embedding_bwd(const float* sigma, const float* W, const float* demb, float* dW, int Bn, int half,
                                            jg_stream_t s) {
  if (!sigma || !W || !demb || !dW || Bn < 1 || half < 1) return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(noise_level_embedding_bwd_kernel, dim3((half + 255) / 256), dim3(256), 0, (hipStream_t)s, sigma, W, demb, dW,
                     Bn, half);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_noise_level_embedding(const float* sigma, const float* W, float* emb, int Bn, int half, jg_stream_t s) {
  if (!sigma || !W || !emb || Bn < 1 || half < 1) return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(noise_level_embedding_kernel, dim3((Bn * 2 * half + 255) / 256), dim3(256), 0, (hipStream_t)s, sigma, W, emb,
                     Bn, half);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_ddpm_p_sample(int dtype, float* y_t, const float* y_cond, const void* noise_hat, const float* z,
                                const float* y_0, const int64_t* mask, const float* coef, void* xin, int B, int C, int H, int W,
                                int Cpad_in, int Cpad_out, int clip_denoised, jg_stream_t s) {
  if (!y_t || !y_cond || !noise_hat || !coef || !xin || Cpad_in < C || Cpad_out < 2 * C || (Cpad_in % 8) || (Cpad_out % 8) ||
      (mask && !y_0))
    return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ddpm_p_sample_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, y_t,
                                              y_cond, (const T*)noise_hat, z, y_0, mask, coef, (T*)xin, B, C, H * W, Cpad_in,
                                              Cpad_out, clip_denoised););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_act_fwd(int dtype, const void* x, void* y, int64_t n, int act, jg_stream_t s) {
  if (!x || !y || n < 8 || n % 8) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((act_fwd_kernel<T>), dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                              (T*)y, (long)(n / 8), act););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_act_bwd(int dtype, const void* y, const void* dy, void* dx, int64_t n, int act, jg_stream_t s) {
  if (!y || !dy || !dx || n < 8 || n % 8) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)s, (const T*)y,
                                              (const T*)dy, (T*)dx, (long)(n / 8), act););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_reflect_pad2d(int dtype, const void* x, void* y, int B, int H, int W, int C, int pad, jg_stream_t s) {
  if (!x || !y || C % 8 || pad < 1 || pad >= H || pad >= W) return JG_ERR_BAD_ARG;
  const long total = (long)B * (H + 2 * pad) * (W + 2 * pad) * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((reflect_pad_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)x, (T*)y, B, H, W, C, pad););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_reflect_pad2d_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int pad, jg_stream_t s) {
  if (!dy || !dx || C % 8 || pad < 1 || pad >= H || pad >= W) return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((reflect_pad_bwd_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)dy, (T*)dx, B, H, W, C, pad););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_crop2d(int dtype, const void* x, void* y, int B, int H, int W, int C, int top, int left, int Ho, int Wo, int adjoint,
                         jg_stream_t s) {
  // adjoint = 0: y[B, Ho, Wo, C] = x[B, H, W, C] window; adjoint = 1: y[B, H, W, C] = x[B, Ho, Wo, C] placed at (top, left), zero elsewhere
  if (!x || !y || C % 8 || top < 0 || left < 0 || Ho < 1 || Wo < 1 || top + Ho > H || left + Wo > W) return JG_ERR_BAD_ARG;
  const long total = (long)B * (adjoint ? (long)H * W : (long)Ho * Wo) * (C / 8);
  if (adjoint) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((crop_kernel<T, true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                                (T*)y, B, H, W, C, top, left, Ho, Wo););
  } else {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((crop_kernel<T, false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                                (T*)y, B, H, W, C, top, left, Ho, Wo););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}
namespace {
// Input gradient of a STRIDED convolution onto an image-like input (<= 4 real channels in an 8-channel pixel), gather form (round 6): the
// first convolutions of the SegFormer encoder (7x7 stride 4, 3 -> 32: segformer/backbone.py PatchEmbed) and of the PatchGAN (4x4 stride 2,
// 3 -> 64: discriminators.py:53-60).  One thread per input pixel collects the ceil(R / s) x ceil(S / s) taps that reach it:
//     dx[b][y][x][ci] = sum_{ky = (y + p) mod s, step s} sum_{kx likewise} sum_co dy[b][(y + p - ky) / s][(x + p - kx) / s][co] w[co][ky][kx][ci].
// Rounds 1-5 ran this as a stride-1 R x S convolution over the zero-dilated dy on the MFMA kernels: s^2 x the useful multiply-adds over an
// output of 3 channels padded to 8 (7x7 stride 4 at 256^2, batch 32: 180 us for 0.8 G useful multiply-adds).  Weights: fp32 in LDS,
// [ky][kx][co][4].
template <typename T>
__global__ __launch_bounds__(256) void conv_dgrad_gather_kernel(const T* __restrict__ dy, const T* __restrict__ w16, T* __restrict__ dx, int B, int H, int W,
                                                                int Ho, int Wo, int Cout, int R, int S, int stride, int pad, float alpha) {
  extern __shared__ float sw[];
  for (int i = threadIdx.x; i < R * S * Cout; i += 256) {
    const int rs = i / Cout, co = i % Cout;
    const T* src = w16 + ((long)co * R * S + rs) * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) sw[i * 4 + c] = to_f32(src[c]);
  }
  __syncthreads();
  // phase-major pixel order: the lanes of a wave share (y mod s, x mod s), hence the taps -- the weight reads are LDS broadcasts and the dy reads
  // of consecutive lanes are consecutive output pixels (x-major order: 4 distinct weight addresses per read, 230 us instead of 70 at 7x7 s4)
  const int Hs = (H + stride - 1) / stride, Ws = (W + stride - 1) / stride;
  const long total = (long)B * stride * stride * Hs * Ws;
  for (long ii = (long)blockIdx.x * 256 + threadIdx.x; ii < total; ii += (long)gridDim.x * 256) {
    long t = ii;
    const int X = (int)(t % Ws); t /= Ws;
    const int Y = (int)(t % Hs); t /= Hs;
    const int px = (int)(t % stride); t /= stride;
    const int py = (int)(t % stride);
    const int b = (int)(t / stride);
    const int x = X * stride + px, y = Y * stride + py;
    if (x >= W || y >= H) continue;
    const long i = ((long)b * H + y) * W + x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int ky = (y + pad) % stride; ky < R; ky += stride) {
      const int ny = y + pad - ky;
      if (ny < 0) break;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
      for (int kx = (x + pad) % stride; kx < S; kx += stride) {
        const int nx = x + pad - kx;
        if (nx < 0) break;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        const uint4* p = reinterpret_cast<const uint4*>(dy + (((long)b * Ho + oy) * Wo + ox) * Cout);
        const float4* wp = reinterpret_cast<const float4*>(sw) + (ky * S + kx) * Cout;
        for (int c8 = 0; c8 < Cout / 8; ++c8) {
          float f[8];
          unpack8<T>(p[c8], f);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float4 w4 = wp[c8 * 8 + q];
            a0 += f[q] * w4.x; a1 += f[q] * w4.y; a2 += f[q] * w4.z; a3 += f[q] * w4.w;
          }
        }
      }
    }
    float o[8] = {alpha * a0, alpha * a1, alpha * a2, alpha * a3, 0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8<T>(o);
  }
}
}  // namespace
// dy [B, Ho, Wo, Cout] 16-bit, w16 [Cout][R][S][8] (the arena's forward copy; input channels 4 .. 7 are padding), dx [B, H, W, 8].
extern "C" int jg_conv_dgrad_gather(int dtype, const void* dy, const void* w16, void* dx, int B, int H, int W, int Ho, int Wo, int Cout, int R, int S,
                                    int stride, int pad, float alpha, jg_stream_t s) {
  if (!dy || !w16 || !dx || B < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || Cout < 8 || (Cout % 8) || R < 1 || S < 1 || stride < 1 || pad < 0) return JG_ERR_BAD_ARG;
  const size_t lds = (size_t)R * S * Cout * 4 * sizeof(float);
  if (lds > 64 * 1024) return JG_ERR_UNSUPPORTED;
  const long total = (long)B * H * W;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((conv_dgrad_gather_kernel<T>), dim3(grid_for(total)), dim3(256), lds, (hipStream_t)s, (const T*)dy, (const T*)w16,
                                              (T*)dx, B, H, W, Ho, Wo, Cout, R, S, stride, pad, alpha););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
namespace {
// ---- row-packed 7x7 head with <= 4 output channels (round 6; ops.py head_conv7) ------------------------------------------------------------
// A 7x7 convolution onto 3 image channels (ReflectionPad2d(3) + Conv2d(64, 3, 7) + Tanh, the last layer of every CUT generator:
// resnet_generator.py:247-263, segformer_generator.py:135-140) pads its 3 output channels to an MFMA tile of 32: 10.7 x the useful work.
// Packed: the 7 tap ROWS become output channels of a 1 x 7 convolution, Z[y'][x][ky * 4 + c] = sum_{kx, ci} xpad[y'][x + kx][ci] w[c][ky][kx][ci]
// (28 of 32 columns live, 7 x fewer multiply-adds), and the rows are summed here: out[y][x][c] = act(b[c] + sum_ky Z[y + ky][x][ky * 4 + c]).
// One thread walks a strip of RY output rows of one column: every z pixel (64 bytes, consecutive lanes = consecutive pixels) is read ONCE and
// feeds the seven output rows it belongs to, held in a ring of accumulators whose slots are compile-time constants (rows in chunks of seven).
// (A first version read z[y + ky][x][ky * 4 ..] per output pixel: seven 8-byte reads at a 64-byte lane stride, 7 x the L2 -> L1 traffic, 160 us.)
template <typename T>
__global__ __launch_bounds__(256) void tapsum7_kernel(const T* __restrict__ z, const float* __restrict__ bias, T* __restrict__ out, int B, int H, int W, int act) {
  constexpr int RY = 32;
  const int nstrip = (H + RY - 1) / RY;
  const long total = (long)B * nstrip * W;
  float bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bv[c] = bias ? bias[c] : 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long t = i / W;
    const int st = (int)(t % nstrip), b = (int)(t / nstrip);
    const int y0 = st * RY, ny = min(H, y0 + RY) - y0;      // output rows y0 .. y0 + ny - 1 <- z rows y0 .. y0 + ny + 5
    float acc[7][4];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[k][c] = bv[c];
    const T* zp = z + (((long)b * (H + 6) + y0) * W + x) * 32;
    T* op = out + (((long)b * H + y0) * W + x) * 8;
    for (int base = 0; base < ny + 6; base += 7) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int r = base + j;                               // z row (relative to y0); slot of output row r - ky = (j - ky) mod 7
        if (r < ny + 6) {
          const uint4* q = reinterpret_cast<const uint4*>(zp + (long)r * W * 32);
          const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
          const uint2 taps[7] = {make_uint2(v0.x, v0.y), make_uint2(v0.z, v0.w), make_uint2(v1.x, v1.y), make_uint2(v1.z, v1.w),
                                 make_uint2(v2.x, v2.y), make_uint2(v2.z, v2.w), make_uint2(v3.x, v3.y)};
#pragma unroll
          for (int ky = 0; ky < 7; ++ky) {
            const int yo = r - ky;
            if (yo >= 0 && yo < ny) {
              float f[4];
              unpack4<T>(taps[ky], f);
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[(j - ky + 7) % 7][c] += f[c];
            }
          }
          const int yd = r - 6;                               // this output row has all seven terms
          if (yd >= 0 && yd < ny) {
            float o[8] = {acc[(j + 1) % 7][0], acc[(j + 1) % 7][1], acc[(j + 1) % 7][2], acc[(j + 1) % 7][3], 0.f, 0.f, 0.f, 0.f};
            if (act == JG_ACT_TANH) {
#pragma unroll
              for (int c = 0; c < 4; ++c) o[c] = tanhf(o[c]);
            }
            *reinterpret_cast<uint4*>(op + (long)yd * W * 8) = pack8<T>(o);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[(j + 1) % 7][c] = bv[c];
          }
        }
      }
    }
  }
}
// adjoint: dpre = dout * act'(out); dZ[y'][x][ky * 4 + c] = dpre[y' - ky][x][c] (zero outside), written twice: dz [B][Hz][W][32] (rows >= H + 6
// zero; the weight-gradient kernel's operand, Hz a multiple of its 8-row tile) and dzm [B][H + 6][W + 12][32] with six zero columns on either
// side (the 1 x 7 input-gradient convolution runs over it without padding).  Same walk: a thread owns a column and a strip of RY z rows, reads
// every (dout, out) pixel once into a ring of seven and writes one 64-byte z pixel per row to each destination.
template <typename T>
__global__ __launch_bounds__(256) void tapspread7_kernel(const T* __restrict__ dout, const T* __restrict__ out, T* __restrict__ dz, T* __restrict__ dzm, int B, int H,
                                                         int W, int Hz, int act) {
  constexpr int RY = 8;       // short strips: the launch is a store stream (128 bytes per row and thread), it needs the threads
  const int Wm = W + 12, nstrip = (Hz + RY - 1) / RY;
  const long total = (long)B * nstrip * Wm;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xm = (int)(i % Wm);
    const long t = i / Wm;
    const int st = (int)(t % nstrip), b = (int)(t / nstrip);
    const int s0 = st * RY, nz = min(Hz, s0 + RY) - s0;     // z rows s0 .. s0 + nz - 1 <- dpre rows s0 - 6 .. s0 + nz - 1
    const int x = xm - 6;
    const bool xin = x >= 0 && x < W;
    uint2 ring[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) ring[k] = make_uint2(0u, 0u);
    for (int base = 0; base < nz + 6; base += 7) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int r = base + j;
        if (r < nz + 6) {
          const int yy = s0 - 6 + r;                         // dpre row loaded into slot j; z row yy is complete with it
          uint2 pv = make_uint2(0u, 0u);
          if (xin && yy >= 0 && yy < H) {
            const long o = (((long)b * H + yy) * W + x) * 8;
            float d[4];
            unpack4<T>(*reinterpret_cast<const uint2*>(dout + o), d);
            if (act == JG_ACT_TANH) {
              float v[4];
              unpack4<T>(*reinterpret_cast<const uint2*>(out + o), v);
#pragma unroll
              for (int c = 0; c < 4; ++c) d[c] *= 1.f - v[c] * v[c];
            }
            pv.x = (uint32_t)to_bits<T>(from_f32<T>(d[0])) | ((uint32_t)to_bits<T>(from_f32<T>(d[1])) << 16);
            pv.y = (uint32_t)to_bits<T>(from_f32<T>(d[2])) | ((uint32_t)to_bits<T>(from_f32<T>(d[3])) << 16);
          }
          ring[j] = pv;
          if (r >= 6) {
            // z pixel of row yy: column ky * 4 + c = dpre[yy - ky][c] = ring slot (j - ky) mod 7
            const uint4 q0 = make_uint4(ring[j].x, ring[j].y, ring[(j + 6) % 7].x, ring[(j + 6) % 7].y);
            const uint4 q1 = make_uint4(ring[(j + 5) % 7].x, ring[(j + 5) % 7].y, ring[(j + 4) % 7].x, ring[(j + 4) % 7].y);
            const uint4 q2 = make_uint4(ring[(j + 3) % 7].x, ring[(j + 3) % 7].y, ring[(j + 2) % 7].x, ring[(j + 2) % 7].y);
            const uint4 q3 = make_uint4(ring[(j + 1) % 7].x, ring[(j + 1) % 7].y, 0u, 0u);
            if (yy < H + 6) {
              uint4* pm = reinterpret_cast<uint4*>(dzm + (((long)b * (H + 6) + yy) * Wm + xm) * 32);
              pm[0] = q0; pm[1] = q1; pm[2] = q2; pm[3] = q3;
            }
            if (xin) {
              uint4* pz = reinterpret_cast<uint4*>(dz + (((long)b * Hz + yy) * W + x) * 32);
              pz[0] = q0; pz[1] = q1; pz[2] = q2; pz[3] = q3;
            }
          }
        }
      }
    }
  }
}
}  // namespace
// z [B][H + 6][W][32] (column ky * 4 + c = tap row ky of output channel c), bias fp32 [>= 4] or NULL, out [B][H][W][8] (channels 4 .. 7 zero);
// act JG_ACT_NONE / JG_ACT_TANH
extern "C" int jg_tapsum7(int dtype, const void* z, const float* bias, void* out, int B, int H, int W, int act, jg_stream_t s) {
  if (!z || !out || B < 1 || H < 1 || W < 1 || (act != JG_ACT_NONE && act != JG_ACT_TANH)) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((tapsum7_kernel<T>), dim3(grid_for((long)B * ((H + 31) / 32) * W)), dim3(256), 0, (hipStream_t)s, (const T*)z, bias, (T*)out, B, H,
                                              W, act););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
// dout, out [B][H][W][8]; dz [B][Hz][W][32], dzm [B][H + 6][W + 12][32], both fully written; Hz >= H + 6
extern "C" int jg_tapspread7(int dtype, const void* dout, const void* out, void* dz, void* dzm, int B, int H, int W, int Hz, int act, jg_stream_t s) {
  if (!dout || !dz || !dzm || B < 1 || H < 1 || W < 1 || Hz < H + 6 || (act != JG_ACT_NONE && act != JG_ACT_TANH) || (act == JG_ACT_TANH && !out))
    return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((tapspread7_kernel<T>), dim3(grid_for((long)B * ((Hz + 7) / 8) * (W + 12))), dim3(256), 0, (hipStream_t)s, (const T*)dout,
                                              (const T*)out, (T*)dz, (T*)dzm, B, H, W, Hz, act););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_dilate2d(int dtype, const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int stride, jg_stream_t s) {
  if (!x || !y || C % 8 || stride < 1 || Ho < (H - 1) * stride + 1 || Wo < (W - 1) * stride + 1) return JG_ERR_BAD_ARG;
  const long total = (long)B * Ho * Wo * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dilate_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                              (T*)y, B, H, W, C, Ho, Wo, stride););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_subsample2d(int dtype, const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int stride, jg_stream_t s) {
  if (!x || !y || C % 8 || stride < 1 || H < (Ho - 1) * stride + 1 || W < (Wo - 1) * stride + 1) return JG_ERR_BAD_ARG;
  const long total = (long)B * Ho * Wo * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((subsample_kernel<T>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)x, (T*)y, B, H, W, C, Ho, Wo, stride););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_channel_sum(int dtype, const void* x, int64_t ldx, float* out, int64_t P, int C, float scale, jg_stream_t s) {
  if (!x || !out || P < 1 || C < 1 || ldx < C) return JG_ERR_BAD_ARG;
  if (C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && P >= 4096) {
    const int noct = C / 8, pl = 256 / noct > 0 ? 256 / noct : 1;
    long g8 = (P + (long)pl * 64 - 1) / ((long)pl * 64);      // ~64 pixels per thread
    if (g8 > 1024) g8 = 1024;
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((channel_sum8_kernel<T>), dim3((unsigned)g8), dim3(256), (size_t)pl * C * sizeof(float), (hipStream_t)s,
                                                (const T*)x, (long)ldx, out, (long)P, C, scale););
    JG_CHECK_LAUNCH();
    return JG_OK;
  }
  long gx = (P + 1023) / 1024;
  if (gx > 512) gx = 512;
  dim3 grid((unsigned)gx, (C + 63) / 64);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((channel_sum_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, (const T*)x, (long)ldx,
                                              out, (long)P, C, scale););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_gather_rows_grouped(int dtype, const void* src, int64_t ld, const int64_t* ids, float* dst, int B, int64_t HW, int C, int P,
                                      int G, int per, jg_stream_t s) {
  if (!src || !ids || !dst || B < 1 || P < 1 || C < 1 || ld < C || G < 1 || per < 1) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gather_rows_kernel<T>), dim3(grid_for((long)B * P * C)), dim3(256), 0, (hipStream_t)s,
                                              (const T*)src, (long)ld, ids, dst, B, (long)HW, C, P, G, per););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_scatter_rows_grouped(int dtype, void* dsrc, int64_t ld, const int64_t* ids, const float* ddst, int B, int64_t HW, int C,
                                       int P, int G, int per, jg_stream_t s) {
  if (!dsrc || !ids || !ddst || B < 1 || P < 1 || C < 1 || ld < C || G < 1 || per < 1) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((scatter_rows_kernel<T>), dim3(grid_for((long)B * P * C)), dim3(256), 0, (hipStream_t)s,
                                              (T*)dsrc, (long)ld, ids, ddst, B, (long)HW, C, P, G, per););
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_gather_rows(int dtype, const void* src, int64_t ld, const int64_t* ids, float* dst, int B, int64_t HW, int C, int P,
                              jg_stream_t s) {
  return jg_gather_rows_grouped(dtype, src, ld, ids, dst, B, HW, C, P, 1, B > 0 ? B : 1, s);
}
extern "C" int jg_scatter_rows(int dtype, void* dsrc, int64_t ld, const int64_t* ids, const float* ddst, int B, int64_t HW, int C,
                               int P, jg_stream_t s) {
  return jg_scatter_rows_grouped(dtype, dsrc, ld, ids, ddst, B, HW, C, P, 1, B > 0 ? B : 1, s);
}
extern "C" int jg_l2norm_fwd(const float* x, float* y, float* nrm, int64_t R, int D, float eps, jg_stream_t s) {
  if (!x || !y || !nrm || R < 1 || D < 1) return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)s, x, y, nrm, (long)R, D, eps);
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_l2norm_bwd(const float* y, const float* nrm, const float* dy, float* dx, int64_t R, int D, float eps, jg_stream_t s) {
  if (!y || !nrm || !dy || !dx || R < 1 || D < 1) return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)s, y, nrm, dy, dx, (long)R, D, eps);
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_gan_loss(int dtype, int mode, const void* pred, float target, float* loss, void* dpred, int64_t Npix, int Cpad, float scale,
                           float grad_scale, jg_stream_t s) {
  if (!pred || !loss || Npix < 1 || Cpad < 1 || mode < 0 || mode > 2) return JG_ERR_BAD_ARG;
  const dim3 grid(grid_for(Npix, 256, 256));
  if (mode == 0) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gan_loss_kernel<T, 0>), grid, dim3(256), 0, (hipStream_t)s, (const T*)pred, target, loss, (T*)dpred,
                                                (long)Npix, Cpad, scale, grad_scale););
  } else if (mode == 1) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gan_loss_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)s, (const T*)pred, target, loss, (T*)dpred,
                                                (long)Npix, Cpad, scale, grad_scale););
  } else {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gan_loss_kernel<T, 2>), grid, dim3(256), 0, (hipStream_t)s, (const T*)pred, target, loss, (T*)dpred,
                                                (long)Npix, Cpad, scale, grad_scale););
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_lsgan_loss(int dtype, const void* pred, float target, float* loss, void* dpred, int64_t Npix, int Cpad, float scale,
                             float grad_scale, jg_stream_t s) {
  return jg_gan_loss(dtype, 0, pred, target, loss, dpred, Npix, Cpad, scale, grad_scale, s);
}

extern "C" int jg_ddpm_multiscale_loss(int dtype, const float* noise, const void* noise_hat, const int64_t* mask, const float* w,
                                       float* losses, void* dnh, float* ws, int B, int C, int H, int W, int Cpad, int nlevels, int l1,
                                       int multiscale, float lambda, float grad_scale, jg_stream_t s) {
  if (!noise || !noise_hat || !losses || Cpad < C || Cpad % 8 || nlevels < 1 || nlevels > 8) return JG_ERR_BAD_ARG;
  if (nlevels > 1 && (!ws || !multiscale || (H & ((1 << (nlevels - 1)) - 1)) || (W & ((1 << (nlevels - 1)) - 1)) || H != W))
    return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W;
  hipStream_t st = (hipStream_t)s;
  if (nlevels > 1) {
    JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ms_loss_down_kernel<T>), dim3(loss_grid((long)B * C * (H / 2) * (W / 2), 256, 512), nlevels - 1),
                                                dim3(256), 0, st, noise, (const T*)noise_hat, mask, w, ws, losses, B, C, H, W, Cpad, nlevels,
                                                l1, lambda););
  }
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ms_loss_grad_kernel<T>), dim3(loss_grid(total, 256, 1024)), dim3(256), 0, st, noise,
                                              (const T*)noise_hat, mask, w, ws, losses, (T*)dnh, B, C, H, W, Cpad, nlevels, l1, multiscale,
                                              lambda, grad_scale););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

// ======================================================================================
// Device-side input pipeline: what the reference's DataLoader workers do per image on the CPU --
// crop (data/online_creation.py crop window), horizontal flip, ToTensor (/255) + Normalize((0.5,), (0.5,)) (data/base_dataset.py:513-528),
// ToTensorMask -> int64 (:892-917) and the self-supervised mask fill  A = B (1 - m) + N(0,1) m,  m = (mask != 0)
// (fill_mask_with_random, data/online_creation.py:1366-1376; data/self_supervised_labeled_mask_dataset.py:46-62) -- as ONE pass over
// the uint8 source: 4 bytes read, 32 written per pixel.  The arithmetic keeps torchvision's operation order (x / 255 - 0.5) / 0.5 in
// fp32, so the result is bit-identical to the CPU pipeline.
// ======================================================================================
namespace {
__global__ __launch_bounds__(256) void input_pipeline_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                             const int32_t* __restrict__ win, const float* __restrict__ noise,
                                                             float* __restrict__ A, float* __restrict__ Bimg,
                                                             int64_t* __restrict__ mout, int B, int H, int W, int S) {
  const long total = (long)B * S * S;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % S);
    const long t = i / S;
    const int y = (int)(t % S), b = (int)(t / S);
    const int oy = win[b * 3], ox = win[b * 3 + 1], flip = win[b * 3 + 2];
    const int sx = ox + (flip ? S - 1 - x : x), sy = oy + y;
    const long sp = ((long)b * H + sy) * W + sx;
    const int mv = mask ? (int)mask[sp] : 0;
    mout[i] = (int64_t)mv;
    const float m = mv != 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = __fdiv_rn(__fsub_rn(__fdiv_rn((float)img[sp * 3 + c], 255.0f), 0.5f), 0.5f);
      const long o = (((long)b * 3 + c) * S + y) * S + x;
      Bimg[o] = v;
      A[o] = __fadd_rn(__fmul_rn(v, 1.0f - m), __fmul_rn(noise ? noise[o] : 0.f, m));
    }
  }
}
}  // namespace

extern "C" int jg_input_pipeline(const uint8_t* img, const uint8_t* mask, const int32_t* win, const float* noise, float* A, float* Bimg,
                                 int64_t* mask_out, int B, int H, int W, int S, jg_stream_t s) {
  if (!img || !win || !A || !Bimg || !mask_out || B < 1 || S < 1 || S > H || S > W) return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(input_pipeline_kernel, dim3(grid_for((long)B * S * S)), dim3(256), 0, (hipStream_t)s, img, mask, win, noise, A, Bimg,
                     mask_out, B, H, W, S);
  JG_CHECK_LAUNCH();
  return JG_OK;
}


// ======================================================================================
// Resize of decoded uint8 images on the device (the `load_size` step of the reference's transforms: data/base_dataset.py:441-443
// `transforms.Resize(osize, interpolation=BICUBIC)` on a PIL image = PIL `Image.resize`, and ResizeMask :749-763 = NEAREST for the label mask).
// PIL resamples in two separable passes of FIXED-POINT arithmetic on uint8 (Pillow src/libImaging/Resample.c: coefficients rounded to
// 22 fractional bits, accumulator started at 1 << 21, result clip8(acc >> 22), the horizontal pass rounded to uint8 before the vertical
// one).  The coefficient tables depend only on (in_size, out_size) and are computed on the host in double precision exactly as
// precompute_coeffs / normalize_coeffs_8bpc do (joligen_amd/data_device.py); the passes below are integer MACs: bit-exact with PIL.
//   pass: out[b, y, x, c] = clip8((2^21 + sum_k in[b, y, xmin[x] + k, c] * kk[x][k]) >> 22)      (horizontal; vertical swaps the roles)
// ======================================================================================
namespace {
__device__ __forceinline__ uint8_t jg_clip8(int v) {
  v >>= 22;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
// horizontal: in [B, H, Win, 3] -> out [B, H, Wout, 3];  vertical (VERT): in [B, Hin, W, 3] -> out [B, Hout, W, 3]
template <bool VERT>
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const int32_t* __restrict__ bounds,
                                                          const int32_t* __restrict__ kk, int ksize, int B, int Hin, int Win, int Hout, int Wout) {
  const long total = (long)B * Hout * Wout;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % Wout);
    const long t = i / Wout;
    const int y = (int)(t % Hout), b = (int)(t / Hout);
    const int o = VERT ? y : x;
    const int lo = bounds[o * 2], n = bounds[o * 2 + 1];
    const int32_t* k = kk + (long)o * ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int j = 0; j < n; ++j) {
      const long sp = VERT ? (((long)b * Hin + lo + j) * Win + x) : (((long)b * Hin + y) * Win + lo + j);
      const int c = k[j];
      s0 += (int)in[sp * 3] * c;
      s1 += (int)in[sp * 3 + 1] * c;
      s2 += (int)in[sp * 3 + 2] * c;
    }
    out[i * 3] = jg_clip8(s0);
    out[i * 3 + 1] = jg_clip8(s1);
    out[i * 3 + 2] = jg_clip8(s2);
  }
}
// PIL NEAREST (Geometry.c ImagingScaleAffine): the source index tables come from the host, which reproduces PIL's incremental double
// accumulation of the source coordinate (it differs from floor((dst + 0.5) * in / out) where that product is an integer)
__global__ __launch_bounds__(256) void resize_nearest_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const int32_t* __restrict__ ytab,
                                                                const int32_t* __restrict__ xtab, int B, int Hin, int Win, int Hout, int Wout) {
  const long total = (long)B * Hout * Wout;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % Wout);
    const long t = i / Wout;
    const int y = (int)(t % Hout), b = (int)(t / Hout);
    out[i] = in[((long)b * Hin + ytab[y]) * Win + xtab[x]];
  }
}
}  // namespace

extern "C" int jg_resample_u8(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize, int vertical, int B, int Hin,
                              int Win, int Hout, int Wout, jg_stream_t s) {
  if (!in || !out || !bounds || !kk || ksize < 1 || B < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return JG_ERR_BAD_ARG;
  if ((vertical && Win != Wout) || (!vertical && Hin != Hout)) return JG_ERR_BAD_ARG;
  const long total = (long)B * Hout * Wout;
  if (vertical) hipLaunchKernelGGL((resample_u8_kernel<true>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, in, out, bounds, kk, ksize, B, Hin, Win, Hout, Wout);
  else hipLaunchKernelGGL((resample_u8_kernel<false>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)s, in, out, bounds, kk, ksize, B, Hin, Win, Hout, Wout);
  JG_CHECK_LAUNCH();
  return JG_OK;
}
extern "C" int jg_resize_nearest_u8(const uint8_t* in, uint8_t* out, const int32_t* ytab, const int32_t* xtab, int B, int Hin, int Win, int Hout,
                                    int Wout, jg_stream_t s) {
  if (!in || !out || !ytab || !xtab || B < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return JG_ERR_BAD_ARG;
  hipLaunchKernelGGL(resize_nearest_u8_kernel, dim3(grid_for((long)B * Hout * Wout)), dim3(256), 0, (hipStream_t)s, in, out, ytab, xtab, B, Hin, Win,
                     Hout, Wout);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

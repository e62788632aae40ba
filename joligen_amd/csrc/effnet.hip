// Kernels of the frozen tf_efficientnet_lite0 feature network of the projected discriminator
// (/root/reference/models/modules/projected_d/projector.py:51-59,251-255: timm `tf_efficientnet_lite0`, MBConv blocks = 1x1 expand ->
// depth-wise 3x3 / 5x5 (TF "SAME" padding, stride 1 or 2) -> 1x1 project, BatchNorm in eval mode, ReLU6).
//
// The network is frozen and its BatchNorms run on their running statistics (ProjectedDiscriminator.train() keeps the feature network
// in eval mode, discriminator.py:267-270), so every BatchNorm is a per-channel affine y = s * x + t with s = gamma / sqrt(var + eps),
// t = beta - mean * s.  Two kernel pairs, NHWC 16-bit activations, fp32 arithmetic, 16-byte accesses (C % 8 == 0):
//   dwconv_affine_act   depth-wise k x k convolution (k = 3 | 5, stride 1 | 2, explicit top / left padding: TF SAME is asymmetric at
//                       stride 2) with the affine and ReLU6 fused; its input gradient (the generator phase back-propagates THROUGH
//                       the frozen network) gathers dz = dy * [0 < y < 6] * s over the taps;
//   chan_affine_act     the affine (+ ReLU6) behind the 1x1 convolutions (which run on the MFMA convolution kernels), and its gradient.
// HBM-bound streaming kernels: a thread owns one 8-channel chunk of one pixel.
#include "common.h"

namespace {

inline int effnet_grid(long total, int block = 256, int cap = 256 * 32) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

template <typename T>
__global__ __launch_bounds__(256) void dwconv_affine_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    T* __restrict__ y, int B, int H, int W, int C, int k, int stride, int pad_t,
                                                                    int pad_l, int Ho, int Wo, int act) {
  const int C8 = C >> 3;
  const long total = (long)B * Ho * Wo * C8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    long t = i / C8;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < k; ++r) {
      const int ih = oh * stride - pad_t + r;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int s = 0; s < k; ++s) {
        const int iw = ow * stride - pad_l + s;
        if ((unsigned)iw >= (unsigned)W) continue;
        float xf[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(x + (((long)b * H + ih) * W + iw) * C + c8 * 8), xf);
        const float4 w0 = *reinterpret_cast<const float4*>(w + (long)(r * k + s) * C + c8 * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(w + (long)(r * k + s) * C + c8 * 8 + 4);
        acc[0] += xf[0] * w0.x; acc[1] += xf[1] * w0.y; acc[2] += xf[2] * w0.z; acc[3] += xf[3] * w0.w;
        acc[4] += xf[4] * w1.x; acc[5] += xf[5] * w1.y; acc[6] += xf[6] * w1.z; acc[7] += xf[7] * w1.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float v = acc[q] * scale[c8 * 8 + q] + shift[c8 * 8 + q];
      acc[q] = act ? relu6f(v) : v;
    }
    *reinterpret_cast<uint4*>(y + i * 8) = pack8<T>(acc);
  }
}

// dx[b, ih, iw, c] = sum over taps (r, s) with (ih + pad_t - r) = oh * stride, (iw + pad_l - s) = ow * stride of dz[b, oh, ow, c] * w[r][s][c],
// dz = dy * [0 < y < 6] * scale   (y = the forward OUTPUT: the clamp's pass-through set is read off it)
template <typename T>
__global__ __launch_bounds__(256) void dwconv_affine_act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const float* __restrict__ w,
                                                                    const float* __restrict__ scale, T* __restrict__ dx, int B, int H, int W, int C,
                                                                    int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act) {
  const int C8 = C >> 3;
  const long total = (long)B * H * W * C8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    long t = i / C8;
    const int iw = (int)(t % W);
    t /= W;
    const int ih = (int)(t % H);
    const int b = (int)(t / H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < k; ++r) {
      const int th = ih + pad_t - r;
      if (th < 0 || th % stride) continue;
      const int oh = th / stride;
      if (oh >= Ho) continue;
      for (int s = 0; s < k; ++s) {
        const int tw = iw + pad_l - s;
        if (tw < 0 || tw % stride) continue;
        const int ow = tw / stride;
        if (ow >= Wo) continue;
        const long o = (((long)b * Ho + oh) * Wo + ow) * C + c8 * 8;
        float g[8], yv[8];
        unpack8<T>(*reinterpret_cast<const uint4*>(dy + o), g);
        if (act) unpack8<T>(*reinterpret_cast<const uint4*>(y + o), yv);
        const float4 w0 = *reinterpret_cast<const float4*>(w + (long)(r * k + s) * C + c8 * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(w + (long)(r * k + s) * C + c8 * 8 + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool pass = !act || (yv[q] > 0.f && yv[q] < 6.f);
          acc[q] += pass ? g[q] * wv[q] : 0.f;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] *= scale[c8 * 8 + q];
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8<T>(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void chan_affine_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ y, long P, int C, int act) {
  const int C8 = C >> 3;
  const long total = P * C8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    float v[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(x + i * 8), v);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float u = v[q] * scale[c8 * 8 + q] + shift[c8 * 8 + q];
      v[q] = act ? relu6f(u) : u;
    }
    *reinterpret_cast<uint4*>(y + i * 8) = pack8<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void chan_affine_act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const float* __restrict__ scale,
                                                                  T* __restrict__ dx, long P, int C, int act) {
  const int C8 = C >> 3;
  const long total = P * C8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    float g[8], yv[8];
    unpack8<T>(*reinterpret_cast<const uint4*>(dy + i * 8), g);
    if (act) unpack8<T>(*reinterpret_cast<const uint4*>(y + i * 8), yv);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool pass = !act || (yv[q] > 0.f && yv[q] < 6.f);
      g[q] = pass ? g[q] * scale[c8 * 8 + q] : 0.f;
    }
    *reinterpret_cast<uint4*>(dx + i * 8) = pack8<T>(g);
  }
}

bool dw_args_ok(const void* a, const void* b, const void* c, int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo) {
  if (!a || !b || !c || B < 1 || H < 1 || W < 1 || C < 8 || C % 8) return false;
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2) || pad_t < 0 || pad_l < 0 || pad_t >= k || pad_l >= k || Ho < 1 || Wo < 1) return false;
  // the last window must start inside the padded image
  return (long)(Ho - 1) * stride - pad_t < H && (long)(Wo - 1) * stride - pad_l < W;
}

}  // namespace

extern "C" int jg_dwconv_affine_act_fwd(int dtype, const void* x, const float* w, const float* scale, const float* shift, void* y, int B, int H,
                                        int W, int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, jg_stream_t s) {
  if (!dw_args_ok(x, w, y, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo) || !scale || !shift || (act != 0 && act != 1)) return JG_ERR_BAD_ARG;
  const long total = (long)B * Ho * Wo * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv_affine_act_fwd_kernel<T>), dim3(effnet_grid(total)), dim3(256), 0, (hipStream_t)s, (const T*)x, w,
                                              scale, shift, (T*)y, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, act););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_dwconv_affine_act_bwd(int dtype, const void* dy, const void* y, const float* w, const float* scale, void* dx, int B, int H, int W,
                                        int C, int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, jg_stream_t s) {
  if (!dw_args_ok(dy, w, dx, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo) || !scale || (act != 0 && act != 1) || (act && !y)) return JG_ERR_BAD_ARG;
  const long total = (long)B * H * W * (C / 8);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv_affine_act_bwd_kernel<T>), dim3(effnet_grid(total)), dim3(256), 0, (hipStream_t)s, (const T*)dy,
                                              (const T*)y, w, scale, (T*)dx, B, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, act););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_chan_affine_act_fwd(int dtype, const void* x, const float* scale, const float* shift, void* y, int64_t P, int C, int act,
                                      jg_stream_t s) {
  if (!x || !scale || !shift || !y || P < 1 || C < 8 || C % 8 || (act != 0 && act != 1)) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_affine_act_fwd_kernel<T>), dim3(effnet_grid(P * (C / 8))), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                              scale, shift, (T*)y, (long)P, C, act););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_chan_affine_act_bwd(int dtype, const void* dy, const void* y, const float* scale, void* dx, int64_t P, int C, int act, jg_stream_t s) {
  if (!dy || !scale || !dx || P < 1 || C < 8 || C % 8 || (act != 0 && act != 1) || (act && !y)) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_affine_act_bwd_kernel<T>), dim3(effnet_grid(P * (C / 8))), dim3(256), 0, (hipStream_t)s, (const T*)dy,
                                              (const T*)y, scale, (T*)dx, (long)P, C, act););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

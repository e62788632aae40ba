// Fused multi-tensor AdamW/Adam + EMA + zero_grad over a flat fp32 parameter arena, and the
// refresh of the 16-bit working copies of the conv weights (straight + flipped/transposed for
// the input-gradient convolution).  One launch each per optimizer step.
//
// Algorithmic bytes per parameter (fp32): read p,g,m,v,ema (20 B) + write p,m,v,ema,g (20 B);
// refresh: read p (4 B) + write w16 (+ w16T) (2-4 B).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema, long n, float lr,
                                                        float beta1, float beta2, float eps, float wd, int decoupled,
                                                        float inv_bc1, float inv_sqrt_bc2, float grad_scale,
                                                        float ema_beta, int zero_grad) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float pv = p[i];
    float gv = g[i] * grad_scale;
    if (wd != 0.f) {
      if (decoupled) pv *= 1.0f - lr * wd;
      else gv += wd * pv;
    }
    const float mv = m[i] + (gv - m[i]) * (1.0f - beta1);  // lerp_, as torch.optim's single-tensor path
    const float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;
    const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
    pv -= lr * inv_bc1 * (mv / denom);
    p[i] = pv;
    m[i] = mv;
    v[i] = vv;
    if (ema) ema[i] = pv + ema_beta * (ema[i] - pv);
    if (zero_grad) g[i] = 0.f;
  }
}

__global__ __launch_bounds__(256) void ema_update_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float beta) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pv = p[i];
    ema[i] = pv + beta * (ema[i] - pv);
  }
}

// One block = one 64 (co) x 64 (ci) tile of one filter tap of one layer, staged through LDS so that the fp32
// master is read in full 256-byte rows (ci fastest) and BOTH 16-bit images are written in full rows: the
// straight copy [CoutP][RS][CinP] (ci fastest) and the flipped + transposed one [CinP][RS][CoutP] (co
// fastest).  (A direct gather for the transposed image read 20x the parameter bytes: 4-byte accesses at
// a stride of RS*Cin floats.)
template <typename T>
__global__ __launch_bounds__(256) void refresh_weights_kernel(const float* __restrict__ p, T* __restrict__ w16,
                                                              T* __restrict__ w16T, const int64_t* __restrict__ desc) {
  __shared__ float tile[64][65];
  const int64_t* d = desc + (long)blockIdx.y * 8;
  const long src = d[0], dst = d[1], dstT = d[2];
  const int Cout = (int)d[3], RS = (int)d[4], Cin = (int)d[5], CoutP = (int)d[6], CinP = (int)d[7];
  const int tci = (CinP + 63) / 64, tco = (CoutP + 63) / 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int t0 = blockIdx.x; t0 < tci * tco * RS; t0 += gridDim.x) {   // block-uniform trip count
    int t = t0;
    const int ci0 = (t % tci) * 64;
    t /= tci;
    const int co0 = (t % tco) * 64;
    const int rs = t / tco;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
      const int co = co0 + r, ci = ci0 + tx;
      const float v = (co < Cout && ci < Cin) ? p[src + ((long)co * RS + rs) * Cin + ci] : 0.f;
      tile[r][tx] = v;
      if (co < CoutP && ci < CinP) w16[dst + ((long)co * RS + rs) * CinP + ci] = from_f32<T>(v);
    }
    if (dstT >= 0) {
      __syncthreads();
#pragma unroll 4
      for (int r = ty; r < 64; r += 4) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < CinP && co < CoutP) w16T[dstT + ((long)ci * RS + (RS - 1 - rs)) * CoutP + co] = from_f32<T>(tile[tx][r]);
      }
      __syncthreads();
    }
  }
}

}  // namespace

extern "C" int jg_adamw_ema(float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1,
                            float beta2, float eps, float wd, int decoupled, int step, float grad_scale, float ema_beta,
                            int zero_grad, jg_stream_t s) {
  if (!p || !g || !m || !v || n < 1 || step < 1) return JG_ERR_BAD_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long grid = (n + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, ema, (long)n, lr, beta1,
                     beta2, eps, wd, decoupled, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, ema_beta, zero_grad);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_ema_update(float* ema, const float* p, int64_t n, float beta, jg_stream_t s) {
  if (!ema || !p || n < 1) return JG_ERR_BAD_ARG;
  long grid = (n + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, ema, p, (long)n, beta);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_refresh_weights(int dtype, const float* p, void* w16, void* w16T, const int64_t* desc, int nlayers,
                                  jg_stream_t s) {
  if (!p || !w16 || !desc || nlayers < 1 || nlayers > 65535) return JG_ERR_BAD_ARG;
  dim3 grid(1152, nlayers);   // 64x64 tiles of a 1024 x 512 x 3x3 layer; larger layers loop, smaller ones exit
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((refresh_weights_kernel<T>), grid, dim3(256), 0, (hipStream_t)s, p, (T*)w16,
                                              (T*)w16T, desc););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

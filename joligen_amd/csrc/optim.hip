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
                                                        float ema_beta, int zero_grad, const int* __restrict__ skip,
                                                        int* __restrict__ nskipped) {
  // skip: optional device flag set by jg_grad_nonfinite -- the step is dropped (GradScaler semantics), gradients cleared
  const bool drop = skip && *skip != 0;
  if (drop && nskipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(nskipped, 1);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (drop) {
      if (zero_grad) g[i] = 0.f;
      continue;
    }
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

// train.py:51-62 also offers RAdam and Lion.  One fused pass each, same arena / EMA / zero_grad contract as adamw_ema_kernel.
// `skip` (optional, device): a non-zero value means the gradient held a non-finite value (jg_grad_nonfinite): the step is dropped like
// torch.cuda.amp.GradScaler.step does -- parameters, moments and EMA untouched, gradients cleared.
//   kind 2 = torch.optim.RAdam (coupled weight decay): m, v as Adam; rho_t = rho_inf - 2 t b2^t / (1 - b2^t);
//            rho_t > 5: p -= lr * (m / bc1) * rect * sqrt(bc2) / (sqrt(v) + eps), else p -= lr * m / bc1   (rect, the flag: host side)
//   kind 3 = Lion (util/lion_pytorch.py:60-82): p *= 1 - lr wd; p -= lr sign(b1 m + (1 - b1) g); m = b2 m + (1 - b2) g
template <int KIND>
__global__ __launch_bounds__(256) void optim_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float* __restrict__ ema, long n, float lr, float beta1,
                                                    float beta2, float eps, float wd, float inv_bc1, float radam_scale, float grad_scale,
                                                    float ema_beta, int zero_grad, const int* __restrict__ skip, int* __restrict__ nskipped) {
  const bool drop = skip && *skip != 0;
  if (drop && nskipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(nskipped, 1);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (drop) {
      if (zero_grad) g[i] = 0.f;
      continue;
    }
    float pv = p[i];
    float gv = g[i] * grad_scale;
    if (KIND == 3) {
      pv *= 1.0f - lr * wd;
      const float mo = m[i];
      const float u = mo * beta1 + gv * (1.0f - beta1);
      pv -= lr * (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f));
      m[i] = mo * beta2 + gv * (1.0f - beta2);
    } else {
      if (wd != 0.f) gv += wd * pv;
      const float mv = m[i] + (gv - m[i]) * (1.0f - beta1);
      const float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;
      const float mh = mv * inv_bc1;
      // radam_scale = rect * sqrt(bc2) when the variance is tractable (rho_t > 5), < 0 otherwise
      pv -= radam_scale >= 0.f ? lr * mh * radam_scale / (sqrtf(vv) + eps) : lr * mh;
      m[i] = mv;
      v[i] = vv;
    }
    p[i] = pv;
    if (ema) ema[i] = pv + ema_beta * (ema[i] - pv);
    if (zero_grad) g[i] = 0.f;
  }
}

// any non-finite gradient -> *flag = 1 (the flag is cleared by the caller); fp16 activations with a static loss scale can overflow
__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, long n, int* __restrict__ flag) {
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = g[i];
    bad |= !(fabsf(x) <= 3.0e38f);       // false for inf and NaN
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
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

extern "C" int jg_adamw_ema_skip(float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float wd, int decoupled, int step, float grad_scale, float ema_beta,
                                 int zero_grad, const int* skip, int* nskipped, jg_stream_t s);

extern "C" int jg_adamw_ema(float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1,
                            float beta2, float eps, float wd, int decoupled, int step, float grad_scale, float ema_beta,
                            int zero_grad, jg_stream_t s) {
  return jg_adamw_ema_skip(p, g, m, v, ema, n, lr, beta1, beta2, eps, wd, decoupled, step, grad_scale, ema_beta, zero_grad, nullptr,
                           nullptr, s);
}

extern "C" int jg_adamw_ema_skip(float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float wd, int decoupled, int step, float grad_scale, float ema_beta,
                                 int zero_grad, const int* skip, int* nskipped, jg_stream_t s) {
  if (!p || !g || !m || !v || n < 1 || step < 1) return JG_ERR_BAD_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long grid = (n + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, ema, (long)n, lr, beta1,
                     beta2, eps, wd, decoupled, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, ema_beta, zero_grad, skip, nskipped);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_optim_step(int kind, float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1, float beta2,
                             float eps, float wd, int step, float grad_scale, float ema_beta, int zero_grad, const int* skip,
                             int* nskipped, jg_stream_t s) {
  if (!p || !g || !m || n < 1 || step < 1) return JG_ERR_BAD_ARG;
  long grid = (n + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  if (kind == 0 || kind == 1)
    return jg_adamw_ema_skip(p, g, m, v, ema, n, lr, beta1, beta2, eps, wd, kind == 1, step, grad_scale, ema_beta, zero_grad, skip, nskipped, s);
  if (kind == 2) {
    if (!v) return JG_ERR_BAD_ARG;
    const double b2t = pow((double)beta2, (double)step), bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - b2t;
    const double rho_inf = 2.0 / (1.0 - (double)beta2) - 1.0, rho_t = rho_inf - 2.0 * step * b2t / bc2;
    float rs = -1.f;
    if (rho_t > 5.0) rs = (float)(sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t)) * sqrt(bc2));
    hipLaunchKernelGGL((optim_kernel<2>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, ema, (long)n, lr, beta1, beta2, eps, wd,
                       (float)(1.0 / bc1), rs, grad_scale, ema_beta, zero_grad, skip, nskipped);
  } else if (kind == 3) {
    hipLaunchKernelGGL((optim_kernel<3>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, p, g, m, v, ema, (long)n, lr, beta1, beta2, eps, wd,
                       1.f, 0.f, grad_scale, ema_beta, zero_grad, skip, nskipped);
  } else {
    return JG_ERR_UNSUPPORTED;
  }
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_grad_nonfinite(const float* g, int64_t n, int* flag, jg_stream_t s) {
  if (!g || !flag || n < 1) return JG_ERR_BAD_ARG;
  long grid = (n + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(grad_nonfinite_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)s, g, (long)n, flag);
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

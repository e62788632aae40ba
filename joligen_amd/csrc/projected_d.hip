// Kernels of the projected discriminator (models/modules/projected_d/{discriminator,blocks}.py, models/modules/loss.py:77-84):
//   * spectral normalisation of the mini-discriminators' convolutions (torch.nn.utils.spectral_norm: one power iteration per
//     training forward, W_sn = W / sigma, gradient through sigma with u, v held constant);
//   * the hinge objective of gan_mode "projected".
// All fp32 on the small weight matrices (<= 512 x 4096); W is the arena's PHYSICAL layout [Cout][R][S][Cin] (= matrix [Cout][K],
// K = RS * Cin, k = rs * Cin + ci), while `v` is kept in the REFERENCE's order (weight.view(Cout, -1) of OIHW: k_ref = ci * RS + rs) so
// that the `weight_v` buffer interchanges with reference checkpoints.
#include "common.h"

namespace {

__device__ __forceinline__ int kref(int k, int RS, int Cin) { return (k % Cin) * RS + k / Cin; }

// t[k] += sum over a slice of 16 rows of W[r][k] u[r]   (t zeroed by the caller; grid = column blocks x row slices: a 512 x 4096
// matrix gives 512 blocks instead of 16)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ t,
                                                     int Cout, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 16, r1 = min(Cout, r0 + 16);
  if (k >= K) return;
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc += W[(long)r * K + k] * u[r];
  atomicAdd(t + k, acc);
}
// s[r] = sum_k W[r][k] t[k] / max(|t|, eps);  nrm[1] += |s|^2          (one wave per row; every wave recomputes |t|^2 -- K <= 4096 --
// and the first one publishes it in nrm[0])
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ t, float* __restrict__ s,
                                                    float* __restrict__ nrm, int Cout, int K, float eps) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= Cout) return;
  float tt = 0.f;
  for (int k = lane; k < K; k += 64) tt += t[k] * t[k];
  tt = wave_sum(tt);
  if (r == 0 && lane == 0) nrm[0] = tt;
  const float inv = 1.0f / fmaxf(sqrtf(tt), eps);
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += W[(long)r * K + k] * (t[k] * inv);
  acc = wave_sum(acc);
  if (lane == 0) {
    s[r] = acc;
    atomicAdd(nrm + 1, acc * acc);
  }
}
// v_ref = t / max(|t|, eps) (reference order), u = s / max(|s|, eps), sigma = u . s
__global__ __launch_bounds__(256) void sn_finish_kernel(const float* __restrict__ t, const float* __restrict__ s, const float* __restrict__ nrm,
                                                        float* __restrict__ u, float* __restrict__ v, float* __restrict__ sigma, int Cout,
                                                        int RS, int Cin, float eps) {
  const int K = RS * Cin;
  const float n1 = fmaxf(sqrtf(nrm[0]), eps), n2 = fmaxf(sqrtf(nrm[1]), eps);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < K; i += gridDim.x * 256) v[kref(i, RS, Cin)] = t[i] / n1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Cout; i += gridDim.x * 256) u[i] = s[i] / n2;
  if (blockIdx.x == 0 && threadIdx.x == 0) *sigma = nrm[1] / n2;
}

// 16-bit working copies of W / sigma: straight [CoutP][RS][CinP] and flipped + transposed [CinP][RS][CoutP] (input-gradient convolution)
template <typename T>
__global__ __launch_bounds__(256) void sn_weights_kernel(const float* __restrict__ W, const float* __restrict__ sigma, T* __restrict__ w16,
                                                         T* __restrict__ w16T, int Cout, int RS, int Cin, int CoutP, int CinP) {
  const float inv = 1.0f / *sigma;
  const long n = (long)CoutP * RS * CinP;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ci = (int)(i % CinP);
    const long q = i / CinP;
    const int rs = (int)(q % RS), co = (int)(q / RS);
    const float v = (co < Cout && ci < Cin) ? W[((long)co * RS + rs) * Cin + ci] * inv : 0.f;
    w16[i] = from_f32<T>(v);
    if (w16T) w16T[((long)ci * RS + (RS - 1 - rs)) * CoutP + co] = from_f32<T>(v);
  }
}

// dot += <dWsn, W>
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dot, long n) {
  float acc = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += a[i] * b[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(dot, acc);
}
// g[r][k] += (dWsn[r][k] - (dot / sigma^2) * sigma * u[r] v[k]) / sigma = (dWsn - <dWsn, W_sn> u v^T) / sigma
__global__ __launch_bounds__(256) void sn_fix_kernel(const float* __restrict__ dWsn, const float* __restrict__ u, const float* __restrict__ v,
                                                     const float* __restrict__ sigma, const float* __restrict__ dot, float* __restrict__ g,
                                                     int Cout, int RS, int Cin) {
  const int K = RS * Cin;
  const float sg = *sigma, proj = *dot / sg;
  const long n = (long)Cout * K;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int k = (int)(i % K), r = (int)(i / K);
    g[i] += (dWsn[i] - proj * u[r] * v[kref(k, RS, Cin)]) / sg;
  }
}

// hinge objective of gan_mode "projected" (loss.py:77-84) on an NHWC logit map with `cvalid` valid leading channels out of cpad:
//   mode 0: mean relu(1 - p)   (D, real)     mode 1: mean relu(1 + p)   (D, fake)     mode 2: mean(-p)   (G)
template <typename T>
__global__ __launch_bounds__(256) void hinge_kernel(const T* __restrict__ pred, float* __restrict__ loss, T* __restrict__ dpred, long npix,
                                                    int cpad, int cvalid, int mode, float scale, float grad_scale) {
  const float inv = 1.0f / (float)(npix * cvalid);
  float acc = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npix * cpad; i += (long)gridDim.x * 256) {
    const int c = (int)(i % cpad);
    float d = 0.f;
    if (c < cvalid) {
      const float p = to_f32(pred[i]);
      if (mode == 0) { const float h = 1.f - p; acc += h > 0.f ? h : 0.f; d = h > 0.f ? -1.f : 0.f; }
      else if (mode == 1) { const float h = 1.f + p; acc += h > 0.f ? h : 0.f; d = h > 0.f ? 1.f : 0.f; }
      else { acc -= p; d = -1.f; }
    }
    if (dpred) dpred[i] = from_f32<T>(d * inv * scale * grad_scale);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * inv * scale);
}

inline unsigned grid1(long n, long cap = 4096) {
  long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int jg_spectral_power_iter(const float* W, float* u, float* v, float* sigma, float* ws, int Cout, int RS, int Cin, float eps,
                                      jg_stream_t s) {
  // ws: fp32 workspace of at least RS * Cin + Cout + 2 floats
  if (!W || !u || !v || !sigma || !ws || Cout < 1 || RS < 1 || Cin < 1) return JG_ERR_BAD_ARG;
  const int K = RS * Cin;
  float* t = ws;
  float* sv = ws + K;
  float* nrm = ws + K + Cout;
  if (hipMemsetAsync(ws, 0, (size_t)(K + Cout + 2) * sizeof(float), (hipStream_t)s) != hipSuccess) return JG_ERR_LAUNCH;
  hipLaunchKernelGGL(sn_wtu_kernel, dim3((K + 255) / 256, (Cout + 15) / 16), dim3(256), 0, (hipStream_t)s, W, u, t, Cout, K);
  hipLaunchKernelGGL(sn_wv_kernel, dim3((Cout + 3) / 4), dim3(256), 0, (hipStream_t)s, W, t, sv, nrm, Cout, K, eps);
  hipLaunchKernelGGL(sn_finish_kernel, dim3(grid1(K, 64)), dim3(256), 0, (hipStream_t)s, t, sv, nrm, u, v, sigma, Cout, RS, Cin, eps);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_spectral_weights(int dtype, const float* W, const float* sigma, void* w16, void* w16T, int Cout, int RS, int Cin, int CoutP,
                                   int CinP, jg_stream_t s) {
  if (!W || !sigma || !w16 || CoutP < Cout || CinP < Cin) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sn_weights_kernel<T>), dim3(grid1((long)CoutP * RS * CinP)), dim3(256), 0, (hipStream_t)s, W, sigma,
                                              (T*)w16, (T*)w16T, Cout, RS, Cin, CoutP, CinP););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_spectral_wgrad_fix(const float* dWsn, const float* W, const float* u, const float* v, const float* sigma, float* g, float* ws,
                                     int Cout, int RS, int Cin, jg_stream_t s) {
  if (!dWsn || !W || !u || !v || !sigma || !g || !ws) return JG_ERR_BAD_ARG;
  const long n = (long)Cout * RS * Cin;
  if (hipMemsetAsync(ws, 0, sizeof(float), (hipStream_t)s) != hipSuccess) return JG_ERR_LAUNCH;
  hipLaunchKernelGGL(sn_dot_kernel, dim3(grid1(n, 1024)), dim3(256), 0, (hipStream_t)s, dWsn, W, ws, n);
  hipLaunchKernelGGL(sn_fix_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)s, dWsn, u, v, sigma, ws, g, Cout, RS, Cin);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

extern "C" int jg_hinge_loss(int dtype, const void* pred, float* loss, void* dpred, int64_t npix, int cpad, int cvalid, int mode, float scale,
                             float grad_scale, jg_stream_t s) {
  if (!pred || !loss || npix < 1 || cpad < cvalid || cvalid < 1 || mode < 0 || mode > 2) return JG_ERR_BAD_ARG;
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((hinge_kernel<T>), dim3(grid1(npix * cpad, 1024)), dim3(256), 0, (hipStream_t)s, (const T*)pred, loss,
                                              (T*)dpred, (long)npix, cpad, cvalid, mode, scale, grad_scale););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

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

// ---- the same steps for ALL spectral-norm layers of a discriminator in one launch each (blockIdx.z / .y = layer) -----------------------------
// A projected discriminator has 14 such convolutions and runs three times per CUT iteration: per-layer launches were 14 x (fill + 3 + 1)
// per forward and 14 x (fill + 2) per backward, ~20 us of work each.  The layers do not depend on activations, so one table-driven
// pass ahead of the forward serves them all.
struct SnDesc {
  const float* W;     // fp32 master weight [Cout][RS][Cin] (arena slice)
  float* u;           // weight_u [Cout]
  float* v;           // weight_v [RS * Cin], reference order
  float* g;           // gradient arena slice of W (the fix step adds into it)
  long ws_off;        // floats into fbuf: t [K] | s [Cout] | nrm [2]      (zero at launch)
  long snap_off;      // floats into fbuf: sigma [1] | u snapshot [Cout] | v snapshot [K]
  long w16_off;       // elements into hbuf: w16 [CoutP][RS][CinP] | w16T [CinP][RS][CoutP]
  long dw_off;        // floats into dbuf: dWsn [Cout][K]
  int Cout, RS, Cin, CoutP, CinP, pad_;
};

__global__ __launch_bounds__(256) void sn_wtu_group_kernel(const SnDesc* __restrict__ tab, float* __restrict__ fbuf) {
  const SnDesc d = tab[blockIdx.z];
  const int K = d.RS * d.Cin;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 16, r1 = min(d.Cout, r0 + 16);
  if (k >= K || r0 >= d.Cout) return;
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc += d.W[(long)r * K + k] * d.u[r];
  atomicAdd(fbuf + d.ws_off + k, acc);
}
__global__ __launch_bounds__(256) void sn_wv_group_kernel(const SnDesc* __restrict__ tab, float* __restrict__ fbuf, float eps) {
  const SnDesc d = tab[blockIdx.y];
  const int K = d.RS * d.Cin;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= d.Cout) return;
  const float* t = fbuf + d.ws_off;
  float* sv = fbuf + d.ws_off + K;
  float* nrm = sv + d.Cout;
  float tt = 0.f;
  for (int k = lane; k < K; k += 64) tt += t[k] * t[k];
  tt = wave_sum(tt);
  if (r == 0 && lane == 0) nrm[0] = tt;
  const float inv = 1.0f / fmaxf(sqrtf(tt), eps);
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += d.W[(long)r * K + k] * (t[k] * inv);
  acc = wave_sum(acc);
  if (lane == 0) {
    sv[r] = acc;
    atomicAdd(nrm + 1, acc * acc);
  }
}
__global__ __launch_bounds__(256) void sn_finish_group_kernel(const SnDesc* __restrict__ tab, float* __restrict__ fbuf, float eps) {
  const SnDesc d = tab[blockIdx.y];
  const int K = d.RS * d.Cin;
  const float* t = fbuf + d.ws_off;
  const float* sv = t + K;
  const float* nrm = sv + d.Cout;
  float* snap = fbuf + d.snap_off;
  const float n1 = fmaxf(sqrtf(nrm[0]), eps), n2 = fmaxf(sqrtf(nrm[1]), eps);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < K; i += gridDim.x * 256) {
    const float val = t[i] / n1;
    const int kr = kref(i, d.RS, d.Cin);
    d.v[kr] = val;
    snap[1 + d.Cout + kr] = val;          // this forward's (u, v): the backward of THIS forward reads them, later forwards iterate on
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < d.Cout; i += gridDim.x * 256) {
    const float val = sv[i] / n2;
    d.u[i] = val;
    snap[1 + i] = val;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) snap[0] = nrm[1] / n2;
}
template <typename T>
__global__ __launch_bounds__(256) void sn_weights_group_kernel(const SnDesc* __restrict__ tab, const float* __restrict__ fbuf, T* __restrict__ hbuf) {
  const SnDesc d = tab[blockIdx.y];
  const float inv = 1.0f / fbuf[d.snap_off];
  const long n = (long)d.CoutP * d.RS * d.CinP;
  T* w16 = hbuf + d.w16_off;
  T* w16T = w16 + n;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ci = (int)(i % d.CinP);
    const long q = i / d.CinP;
    const int rs = (int)(q % d.RS), co = (int)(q / d.RS);
    const float v = (co < d.Cout && ci < d.Cin) ? d.W[((long)co * d.RS + rs) * d.Cin + ci] * inv : 0.f;
    w16[i] = from_f32<T>(v);
    w16T[((long)ci * d.RS + (d.RS - 1 - rs)) * d.CoutP + co] = from_f32<T>(v);
  }
}
// backward: dot[l] = <dWsn_l, W_l>, then g_l += (dWsn_l - (dot / sigma) u v^T) / sigma; `mask` bit l = layer l takes part
__global__ __launch_bounds__(256) void sn_dot_group_kernel(const SnDesc* __restrict__ tab, const float* __restrict__ dbuf, float* __restrict__ dots,
                                                           unsigned long long mask) {
  if (!((mask >> blockIdx.y) & 1ull)) return;
  const SnDesc d = tab[blockIdx.y];
  const long n = (long)d.Cout * d.RS * d.Cin;
  const float* a = dbuf + d.dw_off;
  float acc = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += a[i] * d.W[i];
  acc = wave_sum(acc);
  // ONE atomic per block (round 6): the L sums share a cache line, and with one atomic per wave of 256 blocks the 14 layers queued 14 K
  // read-modify-writes on it -- 189 us per launch for 80 MB of reads (profiles/r06_cut_mobile_kernel_stats.md)
  __shared__ float s_w[4];
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dots + blockIdx.y, (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}
__global__ __launch_bounds__(256) void sn_fix_group_kernel(const SnDesc* __restrict__ tab, const float* __restrict__ fbuf, const float* __restrict__ dbuf,
                                                           const float* __restrict__ dots, unsigned long long mask) {
  if (!((mask >> blockIdx.y) & 1ull)) return;
  const SnDesc d = tab[blockIdx.y];
  const int K = d.RS * d.Cin;
  const float* snap = fbuf + d.snap_off;
  const float sg = snap[0], proj = dots[blockIdx.y] / sg;
  const float* u = snap + 1;
  const float* v = snap + 1 + d.Cout;
  const float* dW = dbuf + d.dw_off;
  const long n = (long)d.Cout * K;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int k = (int)(i % K), r = (int)(i / K);
    d.g[i] += (dW[i] - proj * u[r] * v[kref(k, d.RS, d.Cin)]) / sg;
  }
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

// One power iteration + the 16-bit working copies (straight and flipped / transposed) for ALL `L` spectral-norm layers described by `table`
// (device array of L records, layout `jg_sn_desc` of include/jg355.h).  fbuf: fp32 scratch, its first `zero_floats` floats (the ws regions)
// are cleared here; hbuf: 16-bit scratch for the working copies.  max_K / max_Cout / max_w: the largest K = RS Cin, Cout and padded weight
// element count over the layers (grid sizes).
extern "C" int jg_spectral_group_forward(int dtype, const void* table, int L, float* fbuf, int64_t zero_floats, void* hbuf, int max_K, int max_Cout,
                                         int64_t max_w, float eps, jg_stream_t s) {
  if (!table || !fbuf || !hbuf || L < 1 || L > 64 || max_K < 1 || max_Cout < 1 || max_w < 1 || zero_floats < 0) return JG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const SnDesc* tab = (const SnDesc*)table;
  if (zero_floats && hipMemsetAsync(fbuf, 0, (size_t)zero_floats * sizeof(float), st) != hipSuccess) return JG_ERR_LAUNCH;
  hipLaunchKernelGGL(sn_wtu_group_kernel, dim3((max_K + 255) / 256, (max_Cout + 15) / 16, L), dim3(256), 0, st, tab, fbuf);
  hipLaunchKernelGGL(sn_wv_group_kernel, dim3((max_Cout + 3) / 4, L), dim3(256), 0, st, tab, fbuf, eps);
  hipLaunchKernelGGL(sn_finish_group_kernel, dim3(grid1(max_K, 16), L), dim3(256), 0, st, tab, fbuf, eps);
  JG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sn_weights_group_kernel<T>), dim3(grid1(max_w, 256), L), dim3(256), 0, st, tab, (const float*)fbuf,
                                              (T*)hbuf););
  JG_CHECK_LAUNCH();
  return JG_OK;
}

// Backward companion: for every layer l with bit l of `mask` set, g_l += (dWsn_l - <dWsn_l, W_sn,l> u v^T) / sigma with (sigma, u, v) of the forward
// that filled fbuf; dbuf holds the dWsn_l ([Cout][K] fp32 at dw_off) and `dots` (L floats, ZERO at launch) is scratch.
extern "C" int jg_spectral_group_wgrad_fix(const void* table, int L, const float* fbuf, const float* dbuf, float* dots, uint64_t mask, int64_t max_n,
                                           jg_stream_t s) {
  if (!table || !fbuf || !dbuf || !dots || L < 1 || L > 64 || max_n < 1) return JG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const SnDesc* tab = (const SnDesc*)table;
  hipLaunchKernelGGL(sn_dot_group_kernel, dim3(grid1(max_n, 64), L), dim3(256), 0, st, tab, dbuf, dots, (unsigned long long)mask);
  hipLaunchKernelGGL(sn_fix_group_kernel, dim3(grid1(max_n, 512), L), dim3(256), 0, st, tab, fbuf, dbuf, (const float*)dots, (unsigned long long)mask);
  JG_CHECK_LAUNCH();
  return JG_OK;
}

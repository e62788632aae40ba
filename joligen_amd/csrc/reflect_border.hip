// Border terms of the input gradient of  ReflectionPad2d(1) -> Conv2d(3x3, padding 0)  (round 6).
//
// The CUT ResnetBlocks (models/modules/resnet_architecture/resnet_generator.py:247-275 of the reference) pad by reflection and convolve
// without padding.  The input gradient is the adjoint of the reflection applied to the gradient dp over the PADDED (H + 2) x (W + 2) domain:
//     dp[q][r] = sum_{ky,kx} dy[q - ky][r - kx] w[ky][kx],      dx[fold(q)][fold(r)] += dp[q][r],
//     fold(0) = 1, fold(q) = q - 1 for 1 <= q <= H, fold(H + 1) = H - 2.
// Rounds 1-5 computed dp as a full convolution over the padded domain on the im2col kernel (66 x 66 is not a multiple of the halo kernel's
// 16 x 16 tile: 264 us at ~620 TFLOP/s per 256-channel layer of configs[0], a quarter of that step's kernel time) and folded it with
// reflect_pad_bwd.  But dp restricted to its interior 1 <= q <= H, 1 <= r <= W IS the zero-padded ("same") input gradient on the H x W
// domain -- the halo-resident kernel's shape -- and what remains is the one-pixel RING of dp: four one-dimensional three-tap convolutions
// over the first / last row and column of dy,
//     dp[0][r]     = sum_kx dy[0][r - kx]     w[0][kx]        dp[H + 1][r] = sum_kx dy[H - 1][r - kx] w[2][kx]
//     dp[q][0]     = sum_ky dy[q - ky][0]     w[ky][0]        dp[q][W + 1] = sum_ky dy[q - ky][W - 1] w[ky][2]      (1 <= q <= H),
// i.e. per image four GEMMs of (66 positions) x (3 Cout) x (Cin): 0.4 % of the layer's FLOPs.  This kernel computes them on the MFMA and
// adds them into rows 1 / H - 2 and columns 1 / W - 2 of the dx the halo kernel has written.
//
// Two launches.  (1) ring GEMMs: one workgroup (4 waves x 16 input channels) per (image, line, 64 input channels) keeps the line's
// (W + 2) / 16 position tiles in accumulators and walks K = 3 taps x Cout once; operands come straight from global memory in MFMA fragment
// layout -- both have the reduction index (co) contiguous: dy [pixel][co], and the flipped / transposed weight copy
// WT[ci][2 - ky][2 - kx][co] of the arena -- results go to an fp32 workspace [B][4][L + 2][Cin].  (2) fold: one thread per (pixel of the four
// target lines, channel octet) sums the <= 5 ring values that land on its pixel (a line's positions 0 and 2 fold onto the same pixel, the
// corners belong to two lines) and adds them to dx ONCE: no atomics, reproducible.  (A first version did both in one workgroup per (image,
// 64 channels), four lines one after the other: 135 us per launch for 64 latency-bound workgroups -- as long as the kernel it replaced.)
#include "common.h"

namespace {

constexpr int RB_MAXT = 18;     // position tiles of a line: (L + 2 + 15) / 16 with L <= 272

template <typename T, int NT>
__global__ __launch_bounds__(256) void reflect_ring_gemm_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ wT, float* __restrict__ ws,
                                                                int H, int W, int Cout, int Cin, int Lp) {
  const int b = blockIdx.x, line = blockIdx.y, ci0 = blockIdx.z * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kg = lane >> 4;
  const int ci = ci0 + wave * 16 + l16;                    // this lane's B-operand column
  const bool ci_ok = ci < Cin;
  // line 0: dp[0][r]; 1: dp[H + 1][r]; 2: dp[q][0]; 3: dp[q][W + 1]        (positions r / q = 0 .. L + 1 of the padded axis)
  const bool horiz = line < 2;
  const int L = horiz ? W : H;                          // length of the dy line
  const long pstep = horiz ? lddy : (long)W * lddy;     // dy stride between consecutive positions
  const T* src = dy + (long)b * H * W * lddy + (line == 0 ? 0 : line == 1 ? (long)(H - 1) * W * lddy : line == 2 ? 0 : (long)(W - 1) * lddy);
  const int ntile = (L + 2 + 15) >> 4;
  f32x4 acc[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int tap = 0; tap < 3; ++tap) {
    // weight tap of the padded-domain gradient in the flipped / transposed copy: (ky, kx) -> WT[.][2 - ky][2 - kx][.]
    const int ky = line == 0 ? 0 : line == 1 ? 2 : tap, kx = line == 2 ? 0 : line == 3 ? 2 : tap;
    const T* wrow = wT + ((long)(ci_ok ? ci : 0) * 9 + (2 - ky) * 3 + (2 - kx)) * Cout + kg * 8;
    const T* arow[NT];
    bool a_ok[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      const int sp = mt * 16 + l16 - tap;               // dy position feeding (position, tap)
      a_ok[mt] = mt < ntile && sp >= 0 && sp < L;
      arow[mt] = src + (long)(a_ok[mt] ? sp : 0) * pstep + kg * 8;
    }
#pragma unroll 2
    for (int k0 = 0; k0 < Cout; k0 += 32) {
      const uint4 bq = ldg16<T>(wrow + k0, ci_ok);
      uint4 a[NT];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) a[mt] = ldg16<T>(arow[mt] + k0, a_ok[mt]);
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) acc[mt] = Mfma<T>::run(a[mt], bq, acc[mt]);
    }
  }
  // D: row = position = mt * 16 + kg * 4 + q, col = ci
  if (!ci_ok) return;
  float* o = ws + (((long)b * 4 + line) * Lp) * Cin + ci;
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pos = mt * 16 + kg * 4 + q;
      if (pos < L + 2) o[(long)pos * Cin] = acc[mt][q];
    }
}

// one thread per (target pixel of the four lines, 8 channels): rows 1 and H - 2 in full, columns 1 and W - 2 without those two rows
template <typename T>
__global__ __launch_bounds__(256) void reflect_ring_fold_kernel(const float* __restrict__ ws, T* __restrict__ dx, long lddx, int H, int W, int Cin, int Lp,
                                                                float alpha, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int noct = Cin >> 3;
  const int oc = (int)(i % noct);
  long r = i / noct;
  const int npix = 2 * W + 2 * (H - 2);
  const int pi = (int)(r % npix), b = (int)(r / npix);
  int y, x;
  if (pi < W) { y = 1; x = pi; }
  else if (pi < 2 * W) { y = H - 2; x = pi - W; }
  else {
    const int k = pi - 2 * W, side = k / (H - 2), j = k % (H - 2);      // rows other than 1 and H - 2, in order
    y = j == 0 ? 0 : j == H - 3 ? H - 1 : j + 1;                         // 0, 2, 3, .., H - 3, H - 1
    x = side == 0 ? 1 : W - 2;
  }
  const float* wb = ws + (long)b * 4 * Lp * Cin + oc * 8;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = 0.f;
  auto add = [&](int line, int pos) {
    const float4* s4 = reinterpret_cast<const float4*>(wb + ((long)line * Lp + pos) * Cin);
    const float4 u0 = s4[0], u1 = s4[1];
    v[0] += u0.x; v[1] += u0.y; v[2] += u0.z; v[3] += u0.w; v[4] += u1.x; v[5] += u1.y; v[6] += u1.z; v[7] += u1.w;
  };
  // (every pixel sums its lines in this fixed order; H == 3 / W == 3 put both folds on one pixel)
  if (y == 1) { add(0, x + 1); if (x == 1) add(0, 0); if (x == W - 2) add(0, W + 1); }
  if (y == H - 2) { add(1, x + 1); if (x == 1) add(1, 0); if (x == W - 2) add(1, W + 1); }
  if (x == 1) add(2, y + 1);
  if (x == W - 2) add(3, y + 1);
  T* o = dx + (((long)b * H + y) * W + x) * lddx + oc * 8;
  const uint4 cur = *reinterpret_cast<const uint4*>(o);
  float f[8];
  unpack8<T>(cur, f);
#pragma unroll
  for (int q = 0; q < 8; ++q) f[q] += alpha * v[q];
  *reinterpret_cast<uint4*>(o) = pack8<T>(f);
}

}  // namespace

extern "C" int64_t jg_reflect_dgrad_border_ws_floats(int B, int H, int W, int Cin) { return (int64_t)B * 4 * ((H > W ? H : W) + 2) * Cin; }

// dx [B, H, W, Cin] (pixel stride lddx) holds the zero-padded ("same") input gradient of the 3x3 convolution (jg_conv2d_nt on wT with pad 1);
// adds alpha x the ring terms of the reflection's adjoint in place.  wT: [Cin][3][3][Cout], the flipped / transposed 16-bit weight copy.
// ws: jg_reflect_dgrad_border_ws_floats(B, H, W, Cin) floats of scratch.
extern "C" int jg_reflect_dgrad_border(int dtype, const void* dy, int64_t lddy, const void* wT, void* dx, int64_t lddx, float* ws, int B, int H, int W,
                                       int Cout, int Cin, float alpha, jg_stream_t s) {
  if (!dy || !wT || !dx || !ws || B < 1 || B > 65535 || H < 4 || W < 4 || Cout < 32 || (Cout % 32) || Cin < 8 || (Cin % 8) || lddy < Cout ||
      (lddy % 8) || lddx < Cin || (lddx % 8))
    return JG_ERR_BAD_ARG;
  const int L = H > W ? H : W, Lp = L + 2, nt = (Lp + 15) / 16;
  if (nt > RB_MAXT) return JG_ERR_UNSUPPORTED;
  jg_note_kernel("reflect_ring_gemm_kernel");
  hipStream_t st = (hipStream_t)s;
  dim3 grid(B, 4, (Cin + 63) / 64);
  const long total = (long)B * (2 * W + 2 * (H - 2)) * (Cin / 8);
#define JG_RB_LAUNCH(TT, NTT) hipLaunchKernelGGL((reflect_ring_gemm_kernel<TT, NTT>), grid, dim3(256), 0, st, (const TT*)dy, (long)lddy, (const TT*)wT, ws, H, W, Cout, Cin, Lp)
#define JG_RB_BY_NT(TT) do { if (nt <= 2) JG_RB_LAUNCH(TT, 2); else if (nt <= 3) JG_RB_LAUNCH(TT, 3); else if (nt <= 5) JG_RB_LAUNCH(TT, 5); else if (nt <= 9) JG_RB_LAUNCH(TT, 9); else JG_RB_LAUNCH(TT, RB_MAXT); } while (0)
  if (dtype == JG_F16) {
    JG_RB_BY_NT(f16_t);
    hipLaunchKernelGGL((reflect_ring_fold_kernel<f16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, (f16_t*)dx, (long)lddx, H, W, Cin, Lp, alpha, total);
  } else if (dtype == JG_BF16) {
    JG_RB_BY_NT(bf16_t);
    hipLaunchKernelGGL((reflect_ring_fold_kernel<bf16_t>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, (bf16_t*)dx, (long)lddx, H, W, Cin, Lp, alpha, total);
  } else {
    return JG_ERR_BAD_ARG;
  }
#undef JG_RB_BY_NT
#undef JG_RB_LAUNCH
  JG_CHECK_LAUNCH();
  return JG_OK;
}

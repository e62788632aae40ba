// Halo-resident forward / input-gradient convolution with a LARGE kernel (7x7, stride 1) and FEW output channels on gfx950 MFMA: the content /
// output heads of the CUT generators (nn.ReflectionPad2d(3) + nn.Conv2d(64, 27, 7): resnet_generator.py:247-263, attn_network.py:6-54,
// segformer_generator.py) -- 64 -> 27 (32) channels at 256 x 256 x 32 images -- and their input gradient (32 -> 64 over the padded domain).
//
// Why not gemm_nt.hip: as an implicit GEMM the layer is M = 2.1 M pixels x N = 32 x K = 49 * 64; the im2col A operand crosses L2 -> LDS once per
// tap: 13 GB for a 268 MB tensor, 1.12 ms at 366 TFLOP/s (profiles/r05_cut_kernel_stats.md), L2-traffic bound.  Here a workgroup (4 waves)
// owns a 16 x 16 tile of output pixels x all output channels: the (16 + 6)^2-pixel input HALO comes into LDS once per 64 (32)-channel chunk and
// all 49 taps read it at shifted positions; the weights stream through LDS one tap ROW at a time (7 taps x N x chunk, double buffered).
//
// Register-level reuse of the halo fragments: an MFMA tile is a COLUMN of 16 vertically adjacent pixels (lane = pixel row), a wave owns four
// adjacent columns.  For a tap row r the fragments of halo columns c .. c + 9 serve every horizontal tap s (output column j reads halo
// column j + s): 10 column fragments for 4 x 7 uses -- 0.43 LDS fragment reads per MFMA instead of 0.75.  Vertical shifts (the tap row) are
// address offsets.  Bank conflicts of a column fragment (16 rows, same column): the halo image has an ODD row pitch (23 pixels) so that the
// pixel parity alternates with the row, and the 16-byte chunk index is XORed with the row's upper bits -- 16 rows hit 16 distinct 16-byte
// slots of a 256-byte line; the weight image ([tap][co][chunk], rows = output channels) uses the same rule.
//
// Epilogue: alpha * acc + bias, rounded, transposed through LDS (the halo buffer is free by then) into full 16-byte pixel rows.
#include "conv_params.h"

namespace {

constexpr int TS = 16;        // output tile edge

template <int NCH> __device__ __forceinline__ int kx_swz(int row) { return NCH == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// WBUF 2: the next tap row's weights are written into the other LDS buffer while this one is read; WBUF 1: one buffer (two barriers per tap row) --
// for the configurations that fit TWO workgroups per CU that way, where the other workgroup's MFMAs cover the exposed weight staging
// KR (round 6): tap ROWS; KR = 1 with KS = 7 is the 1 x 7 convolution of the row-packed few-channel heads (ops.py head7x7): one weight stage per
// channel chunk; launched with pad 0 only (the callers hand over tensors that carry their margins)
template <typename T, int KS, int CK, int BN, int WBUF, int KR = KS>
__global__ __launch_bounds__(256, ((TS + KR - 1) * (TS + KS) * (CK / 8) + WBUF * KS * BN * (CK / 8)) * 16 * 2 <= 163840 ? 2 : 1) void conv_kxk_halo_kernel(ConvP p, int tw, int th) {
  constexpr int NCH = CK / 8;                       // 16-byte chunks per pixel of a channel chunk
  constexpr int HR = TS + KR - 1, HC = TS + KS - 1, HP = HC + 1;      // halo rows, cols; row pitch in pixels (odd)
  static_assert((HP & 1) == 1 && (NCH == 8 || NCH == 4), "layout");
  constexpr int HALO16 = HR * HP * NCH;             // halo image in 16-byte chunks
  constexpr int WROW16 = KS * BN * NCH;             // one tap row of weights
  constexpr int NT = BN / 16, KSTEPS = CK / 32;
  constexpr int NCOLF = 4 + KS - 1;                 // column fragments per wave and tap row
  constexpr int HLOADS = (HR * HC * NCH + 255) / 256, WLOADS = (WROW16 + 255) / 256;
  static_assert((HALO16 + WBUF * WROW16) * 16 <= 163840 && TS * TS * BN * 2 <= (HALO16 + WBUF * WROW16) * 16, "LDS");
  __shared__ uint4 sm[HALO16 + WBUF * WROW16];
  uint4* sH = sm;
  uint4* sW = sm + HALO16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int t = blockIdx.x;
  const int tx = t % tw, r2 = t / tw, ty = r2 % th, b = r2 / th;
  const int oh0 = ty * TS, ow0 = tx * TS;
  const T* __restrict__ xb = (const T*)p.x + (long)b * p.H * p.W * p.ldx;
  const T* __restrict__ wg = (const T*)p.w;

  f32x4 acc[4][NT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int ci0 = 0; ci0 < p.Cin; ci0 += CK) {
    if (ci0) __syncthreads();                       // the previous chunk's readers are done with both images
    // ---- halo of this channel chunk: all loads in flight, then the LDS writes ---------------------------------------------------
    {
      uint4 hv[HLOADS];
#pragma unroll
      for (int i = 0; i < HLOADS; ++i) {
        const int pos = i * 256 + tid;
        const int px = pos / NCH, c = pos % NCH;
        const int hy = px / HC, hx = px % HC;
        const int ih = oh0 + hy - p.pad, iw = ow0 + hx - p.pad;
        const bool ok = pos < HR * HC * NCH && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        hv[i] = ldg16(xb + (ok ? ((long)ih * p.W + iw) * p.ldx + ci0 + c * 8 : 0), ok);
      }
#pragma unroll
      for (int i = 0; i < HLOADS; ++i) {
        const int pos = i * 256 + tid;
        if (pos < HR * HC * NCH) {
          const int px = pos / NCH, c = pos % NCH;
          const int hy = px / HC, hx = px % HC;
          sH[(hy * HP + hx) * NCH + (c ^ kx_swz<NCH>(hy))] = hv[i];
        }
      }
    }
    uint4 wv[WLOADS];
    auto wload = [&](int r) {
#pragma unroll
      for (int i = 0; i < WLOADS; ++i) {
        const int pos = i * 256 + tid;
        const int c = pos % NCH, q = pos / NCH;      // q = tap * BN + co
        const int co = q % BN, s = q / BN;
        const bool ok = pos < WROW16 && co < p.N;
        wv[i] = ldg16(wg + (ok ? (long)co * p.ldw + (long)(r * KS + s) * p.Cin + ci0 + c * 8 : 0), ok);
      }
    };
    auto wstore = [&](int buf) {
#pragma unroll
      for (int i = 0; i < WLOADS; ++i) {
        const int pos = i * 256 + tid;
        if (pos < WROW16) {
          const int c = pos % NCH, q = pos / NCH;
          sW[buf * WROW16 + q * NCH + (c ^ kx_swz<NCH>(q % BN))] = wv[i];
        }
      }
    };
    wload(0);
    wstore(0);
    __syncthreads();

    for (int r = 0; r < KR; ++r) {
      const bool more = r + 1 < KR;
      if (more) wload(r + 1);
      // column fragments of halo rows r .. r + 15: lane (pixel row l15, channel group g)
      uint4 fa[NCOLF][KSTEPS];
      const int hy = l15 + r;
#pragma unroll
      for (int j = 0; j < NCOLF; ++j)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
          fa[j][ks] = sH[(hy * HP + wave * 4 + j) * NCH + ((ks * 4 + g) ^ kx_swz<NCH>(hy))];
      const uint4* wb = sW + (WBUF == 2 ? (r & 1) : 0) * WROW16;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        uint4 fb[NT][KSTEPS];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            const int co = n * 16 + l15;
            fb[n][ks] = wb[(s * BN + co) * NCH + ((ks * 4 + g) ^ kx_swz<NCH>(co))];
          }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) acc[j][n] = Mfma<T>::run(fa[j + s][ks], fb[n][ks], acc[j][n]);
      }
      if (WBUF == 1) __syncthreads();           // every wave is done reading the only buffer
      if (more) wstore(WBUF == 2 ? ((r + 1) & 1) : 0);
      __syncthreads();
    }
  }

  // ---- epilogue: D row = pixel row g * 4 + q, col = output channel l15 -> [pixel][channel] rows in LDS -> 16-byte stores -------------
  T* so = reinterpret_cast<T*>(sm);           // both images are free: the last tap row ended with a barrier
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = n * 16 + l15;
    const float bv = (p.bias && co < p.N) ? p.bias[co] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) so[((g * 4 + q) * TS + wave * 4 + j) * BN + co] = from_f32<T>(p.alpha * acc[j][n][q] + bv);
  }
  __syncthreads();
  T* __restrict__ yb = (T*)p.y + (long)b * p.Ho * p.Wo * p.ldy;
  constexpr int OC = BN / 8;
#pragma unroll
  for (int i = 0; i < TS * TS * OC / 256; ++i) {
    const int id = i * 256 + tid;
    const int px = id / OC, cc = id % OC;
    const int oh = oh0 + px / TS, ow = ow0 + px % TS;
    if (oh < p.Ho && ow < p.Wo && cc * 8 < p.N)
      *reinterpret_cast<uint4*>(yb + ((long)oh * p.Wo + ow) * p.ldy + cc * 8) = reinterpret_cast<const uint4*>(so)[px * OC + cc];
  }
}

template <typename T>
bool launch_kxk(const ConvP& p, hipStream_t st) {
  const int tw = (p.Wo + TS - 1) / TS, th = (p.Ho + TS - 1) / TS;
  const dim3 grid((unsigned)(p.B * tw * th));
  if (p.R == 1) {      // 1 x 7 stages of the row-packed heads: forward (64 -> 28 (32) packed channels), input gradient (28 (32) packed -> 64)
    if (p.N <= 32 && p.Cin % 32 == 0 && jg_tune(JG_TUNE_CONV_KXK) != 2) hipLaunchKernelGGL((conv_kxk_halo_kernel<T, 7, 32, 32, 1, 1>), grid, dim3(256), 0, st, p, tw, th);
    else if (p.N <= 32 && p.Cin % 64 == 0) hipLaunchKernelGGL((conv_kxk_halo_kernel<T, 7, 64, 32, 1, 1>), grid, dim3(256), 0, st, p, tw, th);
    else if (p.N <= 64 && p.Cin % 32 == 0) hipLaunchKernelGGL((conv_kxk_halo_kernel<T, 7, 32, 64, 1, 1>), grid, dim3(256), 0, st, p, tw, th);
    else return false;
    return true;
  }
  if (p.R == 3) {      // 3x3 with at most 32 output channels (the UNet's 64 -> 3 (8) output convolution at 256 x 256): 68 KB, two workgroups per CU
    if (p.N > 32 || p.Cin % 64) return false;
    hipLaunchKernelGGL((conv_kxk_halo_kernel<T, 3, 64, 32, 2>), grid, dim3(256), 0, st, p, tw, th);
    return true;
  }
  const int mode = jg_tune(JG_TUNE_CONV_KXK);      // 1: two workgroups per CU wherever a configuration allows it; 2: the one-workgroup forms (A/B)
  if (p.N <= 32 && p.Cin % 64 == 0 && mode == 2) {
    hipLaunchKernelGGL((conv_kxk_halo_kernel<T, 7, 64, 32, 2>), grid, dim3(256), 0, st, p, tw, th);
  } else if (p.N <= 32 && p.Cin % 32 == 0) {
    hipLaunchKernelGGL((conv_kxk_halo_kernel<T, 7, 32, 32, 2>), grid, dim3(256), 0, st, p, tw, th);
  } else if (p.N <= 64 && p.Cin % 32 == 0 && mode == 2) {
    hipLaunchKernelGGL((conv_kxk_halo_kernel<T, 7, 32, 64, 2>), grid, dim3(256), 0, st, p, tw, th);
  } else if (p.N <= 64 && p.Cin % 32 == 0) {
    hipLaunchKernelGGL((conv_kxk_halo_kernel<T, 7, 32, 64, 1>), grid, dim3(256), 0, st, p, tw, th);
  } else {
    return false;
  }
  return true;
}

}  // namespace

bool jg_conv_kxk_try(int dtype, const ConvP& p, int nbatch, hipStream_t st) {
  if (!jg_tune(JG_TUNE_CONV_KXK)) return false;
  const bool row1 = p.R == 1 && p.S == 7 && p.pad == 0;
  if (nbatch != 1 || (!row1 && ((p.R != 7 && p.R != 3) || p.S != p.R)) || p.stride != 1 || p.out_f32 || p.res || p.stats || p.reflect || p.x_up || p.y_pool || p.res_up) return false;
  if (p.N > 64 || (p.N & 7) || (p.Cin & 31) || (p.ldy & 7) || (p.ldx & 7) || (p.ldw & 7)) return false;
  if ((long)p.B * p.Ho * p.Wo < 16384) return false;          // tiny launches: the generic kernel's split-K forms serve them
  if (p.R == 3 && (p.N > 32 || (long)p.B * p.Ho * p.Wo < 262144)) return false;   // 3x3: only the few-channel layers the 64-wide halo kernels do not serve
  if ((long)p.B * p.H * p.W * p.ldx >= (1L << 31) || (long)p.B * p.Ho * p.Wo * p.ldy >= (1L << 31)) return false;
  bool ok = false;
  if (dtype == JG_F16) ok = launch_kxk<f16_t>(p, st);
  else if (dtype == JG_BF16) ok = launch_kxk<bf16_t>(p, st);
  if (ok) jg_note_kernel(p.R == 3 ? "conv_kxk_halo_kernel<3x3>" : p.R == 1 ? "conv_kxk_halo_kernel<1x7>" : "conv_kxk_halo_kernel<7x7>");
  return ok;
}

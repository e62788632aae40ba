// Streaming 1x1 convolution for the wide, shallow layers of the UNet (skip / shortcut convs at full resolution: 64 -> 128,
// 64 -> 192, 128 -> 64 ... channels on 32 x 256^2 pixels).  These are HBM-bound (a few hundred FLOP per byte at most) and the
// generic LDS-DMA implicit-GEMM kernel reaches only ~2 TB/s on them: one K step per tile leaves its pipeline nothing to overlap.
//
// Here nothing goes through LDS: MFMA computes D^T = W . X^T per 16-pixel tile,
//   A operand = weights (row = output channel), loaded ONCE per wave into registers for the whole kernel;
//   B operand = pixels  (col = pixel): lane (l & 15, l >> 4) loads 16 contiguous bytes of pixel row (l & 15) -- a wave instruction
//               covers 16 full 64-byte segments, no im2col, no barrier;
//   D         = [16 output channels] x [16 pixels]: a lane owns 4 consecutive channels of ONE pixel; the weight rows of two MFMA tiles
//               are interleaved so that a lane's 2 x 4 channels are 8 consecutive ones -> one 16-byte store (and one 16-byte read of
//               the residual) per lane and tile pair.
// A wave walks pixel tiles with a stride of the total wave count and keeps the fragments of the next tile in flight.
//   y = alpha * conv + bias + res_scale * res     (same epilogue contract as the other convolution kernels)
#include <stdlib.h>

#include "conv_params.h"

namespace {

template <typename T, int PAIRS, int KS>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(ConvP p, int ntiles) {
  const int lane = threadIdx.x & 63;
  const int a = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * (PAIRS * 32);
  const T* __restrict__ x = (const T*)p.x;
  const T* __restrict__ w = (const T*)p.w;
  const T* __restrict__ res = (const T*)p.res;
  T* __restrict__ y = (T*)p.y;

  // weights: tile (pair, half) row a  <->  output channel n0 + pair*32 + (a >> 2)*8 + half*4 + (a & 3)
  uint4 wf[PAIRS][2][KS];
#pragma unroll
  for (int pr = 0; pr < PAIRS; ++pr)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int n = n0 + pr * 32 + (a >> 2) * 8 + hf * 4 + (a & 3);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wf[pr][hf][ks] = *reinterpret_cast<const uint4*>(w + (long)n * p.ldw + ks * 32 + g * 8);
    }
  // bias of this lane's 8 channels per pair: n0 + pair*32 + g*8 + 0..7
  float bs[PAIRS][8];
#pragma unroll
  for (int pr = 0; pr < PAIRS; ++pr)
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[pr][j] = p.bias ? p.bias[n0 + pr * 32 + g * 8 + j] : 0.f;

  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  auto load_tile = [&](int t, uint4* xf) {
    const T* px = x + ((long)t * 16 + a) * p.ldx + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const uint4*>(px + ks * 32);
  };
  // PF tiles of input fragments in flight per wave (a memory-bound stream needs ~64 KB in flight per CU)
  constexpr int PF = KS <= 2 ? 4 : (KS <= 4 ? 2 : 1);
  uint4 ring[PF][KS];
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (wave + s * nwaves < ntiles) load_tile(wave + s * nwaves, ring[s]);
  for (int base = wave; base < ntiles; base += PF * nwaves) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int t = base + s * nwaves;
      if (t >= ntiles) break;
      const long prow = ((long)t * 16 + a);
      uint4 rf[PAIRS];
      if (res) {
#pragma unroll
        for (int pr = 0; pr < PAIRS; ++pr) rf[pr] = *reinterpret_cast<const uint4*>(res + prow * p.ldres + n0 + pr * 32 + g * 8);
      }
      f32x4 acc[PAIRS][2];
#pragma unroll
      for (int pr = 0; pr < PAIRS; ++pr) {
        acc[pr][0] = acc[pr][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          acc[pr][0] = Mfma<T>::run(wf[pr][0][ks], ring[s][ks], acc[pr][0]);
          acc[pr][1] = Mfma<T>::run(wf[pr][1][ks], ring[s][ks], acc[pr][1]);
        }
      }
      const int tn = t + PF * nwaves;       // refill this slot as soon as its fragments have been consumed
      if (tn < ntiles) load_tile(tn, ring[s]);
#pragma unroll
      for (int pr = 0; pr < PAIRS; ++pr) {
        float o[8] = {acc[pr][0][0], acc[pr][0][1], acc[pr][0][2], acc[pr][0][3], acc[pr][1][0], acc[pr][1][1], acc[pr][1][2], acc[pr][1][3]};
        float r8[8];
        if (res) unpack8<T>(rf[pr], r8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = p.alpha * o[j] + bs[pr][j] + (res ? p.res_scale * r8[j] : 0.f);
        *reinterpret_cast<uint4*>(y + prow * p.ldy + n0 + pr * 32 + g * 8) = pack8<T>(o);
      }
    }
  }
}

template <typename T, int PAIRS, int KS>
void launch_1x1(const ConvP& p, hipStream_t st) {
  const int ntiles = p.M / 16;
  const int ngroups = p.N / (PAIRS * 32);
  int bx = (ntiles + 3) / 4;                 // one tile per wave at least
  const int cap = 256 * 8 / ngroups > 64 ? 256 * 8 / ngroups : 64;
  if (bx > cap) bx = cap;
  hipLaunchKernelGGL((conv1x1_stream_kernel<T, PAIRS, KS>), dim3(bx, ngroups), dim3(256), 0, st, p, ntiles);
}

template <typename T>
bool dispatch_1x1(const ConvP& p, hipStream_t st) {
  const int ks = p.Cin / 32;
  const bool wide = p.N % 128 == 0;
  switch (ks) {
    case 2: wide ? launch_1x1<T, 4, 2>(p, st) : launch_1x1<T, 2, 2>(p, st); return true;
    case 4: launch_1x1<T, 2, 4>(p, st); return true;      // <4, 4> would need 256+ VGPRs
    case 6: launch_1x1<T, 2, 6>(p, st); return true;
    case 8: launch_1x1<T, 2, 8>(p, st); return true;
    default: return false;
  }
}

}  // namespace

// true when the shape was handled here (gemm_nt.hip falls through to the generic kernel otherwise)
bool jg_conv1x1_try(int dtype, const ConvP& p, int nbatch, hipStream_t st) {
  static const int off = [] { const char* e = getenv("JG_CONV1X1"); return e && atoi(e) == 0; }();
  if (off) return false;
  if (nbatch != 1 || p.nh != 1 || p.R != 1 || p.S != 1 || p.pad != 0 || p.stride != 1 || p.out_f32 || p.stats || p.reflect) return false;
  if (p.Cin % 32 || p.Cin > 256 || p.N % 64 || (p.M & 15)) return false;
  if (p.ldx % 8 || p.ldy % 8 || p.ldw % 8 || (p.res && p.ldres % 8)) return false;
  // only where the layer is memory-bound: few hundred FLOP per byte; deep / low-resolution layers stay on the MFMA-tiled GEMM
  if ((long)p.M < 65536) return false;
  if (dtype == JG_F16) return dispatch_1x1<f16_t>(p, st);
  if (dtype == JG_BF16) return dispatch_1x1<bf16_t>(p, st);
  return false;
}

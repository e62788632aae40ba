// Kernel-side parameter block shared by the convolution kernels (gemm_nt.hip, conv_halo.hip).
#pragma once
// mirror index of nn.ReflectionPad2d for a 1-pixel border: -1 -> 1, n -> n - 2
#define JG_REFLECT1(i, n) ((i) < 0 ? -(i) : ((i) >= (n) ? 2 * ((n) - 1) - (i) : (i)))
#include "common.h"

struct ConvP {
  const char* x; const char* w; char* y; const float* bias; const char* res;
  int M, N, K;
  int H, W, Cin, R, S, pad, stride, Ho, Wo;
  long ldx, ldw, ldy, ldres;
  int nh;
  long sxb, sxh, swb, swh, syb, syh, srb, srh;
  float alpha, res_scale;
  int out_f32;
  int B;
  float* stats;   // optional fp32 (sum, sum of squares) per (image, output channel) of the output, accumulated
  long ldstats;   //   atomically at stats[((b * nslots + slot) * ldstats + n) * 2 + {0,1}] (GroupNorm statistics of the consumer);
  int nslots;     //   slot = tile index % nslots spreads the same-address atomic chains (the consumer sums the slots)
  // stats_mode 1: the epilogue accumulates the GroupNorm-BACKWARD reductions of the norm whose output
  // gradient this convolution produces (y = dL/d act(a*x+b)):  (sum du, sum du*x), du = y * act'(a*gx + b),
  // gx = the norm's input (pixel stride gldx), gab = its [B][N][2] (a, b) coefficients.
  int stats_mode;
  int dbg;        // ablation switches of the dev tools (JG_HALO_DBG): 1 = no epilogue, 2 = no MFMA/LDS reads, 4 = no halo DMA
  const char* gx; long gldx; const float* gab; int gact;
  int y_pool;     // 1: y is [B, Ho/2, Wo/2, N] = 2x2 SUM-pool of alpha * conv (halo kernel only; no bias / residual / statistics)
  int x_up;       // 1: x is [B, H/2, W/2, Cin] and is read through the nearest-upsample index map (conv over Upsample(x), halo kernel only);
                  // 2: the same convolution in its sub-pixel form (four 2x2-tap phases, folded weights)
  int res_up;     // 1: res is [B, Ho/2, Wo/2, N] and is read through the nearest-upsample index map (UNet up-block skip path)
  int reflect;    // 1: out-of-image halo pixels mirror the interior (nn.ReflectionPad2d(1) in front of a pad-0 3x3 conv)
  // split-K of the generic LDS-DMA kernel (gemm_nt.hip): splitk > 1 = blockIdx.y owns a slice of the K loop and stores its raw fp32
  // partial tile to ws[((split * nbatch + z) * M + m) * N + n]; jg_splitk_finalize sums the slices in a fixed order (deterministic)
  float* ws; int splitk;
  // streaming 1x1 kernel only (jg_conv1x1_gn_apply): ALSO write ay[m][c] = act(aab[b][c][0] * x[m][c] + aab[b][c][1]) for the Cin input
  // channels -- the GroupNorm apply pass of the tensor this convolution reads (ResBlock: skip_connection(x) next to act(norm(x)))
  const float* aab; char* ay; long lday; int aact;
  // streaming 1x1 kernel only (jg_conv1x1_gn_bwd_apply): the epilogue ADDS the GroupNorm-backward apply step of the tensor whose gradient
  // this convolution produces: y += du P + gx Q + R (+ sc1 add1 + sc2 add2), du = gdy act'(a gx + b); bab [B][N][2], bpqr [B][N][3]
  const char* bgx; long bldgx; const char* bgdy; long bldgdy; const float* bab; const float* bpqr;
  const char* badd1; long bldadd1; float bsc1; const char* badd2; long bldadd2; float bsc2; int bact;
};

// output pixel row m = (b * Ho + oh) * Wo + ow  ->  row of the half-resolution residual
__device__ __forceinline__ long jg_res_up_row(const ConvP& p, long m) {
  const int ow = (int)(m % p.Wo);
  const long t = m / p.Wo;
  const int oh = (int)(t % p.Ho);
  const long b = t / p.Ho;
  return (b * (p.Ho >> 1) + (oh >> 1)) * (p.Wo >> 1) + (ow >> 1);
}

// Per-wave reduction of the epilogue's (sum, sum^2) partials over the 16 pixel lanes of an MFMA tile
// column group, then one atomic pair per channel from lane (l & 15) == 0.
__device__ __forceinline__ void jg_stats_flush(float* stats, long row, int n, const float* s1, const float* s2, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a = s1[q], b = s2[q];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
    }
    if ((lane & 15) == 0) {
      atomicAdd(stats + (row + n + q) * 2, a);
      atomicAdd(stats + (row + n + q) * 2 + 1, b);
    }
  }
}

// conv_halo.hip: returns true when the shape was handled by the halo-resident 3x3 kernel.
bool jg_conv_halo_try(int dtype, const ConvP& p, int nbatch, hipStream_t st);
// conv_p64.hip: returns true when the shape was handled by the persistent Cin == 64 kernel (weights resident in LDS).
bool jg_conv_p64_try(int dtype, const ConvP& p, int nbatch, hipStream_t st);
// conv1x1.hip: returns true when the shape was handled by the streaming (LDS-free) 1x1 kernel.
bool jg_conv1x1_try(int dtype, const ConvP& p, int nbatch, hipStream_t st);
// conv_kxk.hip: returns true when the shape was handled by the halo-resident 7x7 kernel (few output channels, stride 1).
bool jg_conv_kxk_try(int dtype, const ConvP& p, int nbatch, hipStream_t st);

// Kernel-side parameter block shared by the convolution kernels (gemm_nt.hip, conv_halo.hip).
#pragma once
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
  float* stats;   // optional [B][N][2] fp32 (sum, sum of squares) of the stored output, accumulated atomically
};

// conv_halo.hip: returns true when the shape was handled by the halo-resident 3x3 kernel.
bool jg_conv_halo_try(int dtype, const ConvP& p, int nbatch, hipStream_t st);

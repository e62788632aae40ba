// Kernel-side parameter block shared by the weight-gradient kernels (gemm_tn.hip, wgrad_halo.hip).
#pragma once
#include "common.h"

struct WgP {
  const char* dy; const char* x; char* dw; float* dbias;
  int Mpix, Cout, Ktot;  // reduction length, rows, cols (= R*S*Cin)
  int H, W, Cin, R, S, pad, stride, Ho, Wo;
  int Cin_out, Cout_out;
  long lddy, ldx, lddw;
  int nh, splitk;
  long sdyb, sdyh, sxb, sxh, sdwb, sdwh;
  float alpha;
  int out_mode;
  int B;
  float dbias_scale;
  int x_up;      // 1: x is [B, H/2, W/2, Cin], read through the nearest-upsample index map (halo kernel only)
  int reflect;   // 1: the x halo mirrors the interior at the image border (ReflectionPad2d(1) + pad-0 3x3 conv)
};

// wgrad_halo.hip: returns true when the shape was handled by the halo-resident 3x3 kernel.
// dry_run: only answer whether the kernel WOULD take the problem (the grouped entry validates every descriptor before it launches anything)
bool jg_wgrad_halo_try(int dtype, const WgP& p, int nbatch, hipStream_t st, bool dry_run = false);
// wgrad_kxk.hip: returns true when the shape was handled by the halo-resident large-kernel (7x7) weight-gradient kernel.
bool jg_wgrad_kxk_try(int dtype, const WgP& p, int nbatch, hipStream_t st, bool dry_run = false);
// wgrad_sw.hip: sliding-window form of the 16-row x 64-co tile (v_mfma_f32_32x32x16, one pixel row per K step); the caller has validated the shape.
void jg_wgrad_sw_launch(int dtype, const WgP& p, hipStream_t st);

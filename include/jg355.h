/* jg355.h -- C ABI of libjg355.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * joliGEN training-step hot path.
 *
 * Boundary rules (the reference's own precedent for a native op is
 * models/modules/op/upfirdn2d.cpp:8-30 + upfirdn2d.py:19-167):
 *   - plain pointers and sizes only, no torch types;
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator);
 *   - the caller passes the HIP stream; no internal synchronisation, no hidden allocation;
 *   - return 0 on success, a negative JG_ERR_* code otherwise; never throws;
 *   - thread-safe with respect to distinct streams.
 * Activations are 16-bit (JG_F16 / JG_BF16) NHWC with C % 8 == 0; statistics, biases,
 * norm affine parameters, master weights, gradients of parameters are fp32.
 *
 * Each entry point names the reference call it replaces (paths relative to /root/reference).
 */
#ifndef JG355_H
#define JG355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* jg_stream_t; /* hipStream_t */

enum { JG_F16 = 0, JG_BF16 = 1 };
enum { JG_OK = 0, JG_ERR_BAD_ARG = -1, JG_ERR_UNSUPPORTED = -2, JG_ERR_LAUNCH = -3 };
enum { JG_ACT_NONE = 0, JG_ACT_SILU = 1, JG_ACT_RELU = 2, JG_ACT_LRELU = 3 /* slope 0.2 */, JG_ACT_TANH = 4 /* jg_act_* only */ };
enum { JG_OUT_ATOMIC_F32 = 0, JG_OUT_STORE_F32 = 1, JG_OUT_STORE_T = 2 };

int jg_version(void);
const char* jg_strerror(int code);

/* Dispatch switches.  Each switch is read once from the environment variable of the same name
 * ("JG_HALO_CFG", "JG_WGRAD_HALO_CFG", "JG_CONV_VARIANT", "JG_WGRAD_VARIANT", "JG_SINKHORN_GENERIC", "JG_CONV1X1",
 * "JG_GN_REVERSE", "JG_HALO_DBG", "JG_PERSIST64", "JG_HALO_PIPE", "JG_WGRAD_PIPE", "JG_CONV_SPLITK", "JG_CONV_SMALL_TILE",
 * "JG_GN_FUSED", "JG_GN_FUSED_CAP", "JG_GN_FUSED_DBG", "JG_GN_FUSED_SLEEP", "JG_WGRAD_LDS_PAD", and the grid shapes of the SegFormer
 * backward kernels "JG_LN_BWD_CAP" (256 workgroups), "JG_DW_BWD_CAP" (512) and "JG_DW_BWD_PPT" (8 pixels per thread));
 * jg_set_tuning overrides it for the rest of the process (parity tests use
 * it to force a tile configuration that the automatic choice only takes at bench-sized grids).  No reference counterpart:
 * the reference delegates kernel choice to cuDNN's heuristics (torch.backends.cudnn.benchmark, train.py:38-48).
 * Returns JG_OK / JG_ERR_BAD_ARG (unknown name); jg_get_tuning returns the current value or -1. */
int jg_set_tuning(const char* name, int value);
int jg_get_tuning(const char* name);
/* name of the kernel instance the calling thread's last jg_conv2d_nt / jg_conv2d_wgrad_tn launch was dispatched to ("" when the dispatch
 * site does not record one): profiling tools label their rows with what ran */
const char* jg_last_kernel(void);

/* Implicit-GEMM convolution / batched GEMM "NT" on MFMA (v_mfma_f32_16x16x32_{f16,bf16}).
 *   y[z][m][n] = alpha * sum_k A[z][m][k] * w[z][n][k] + bias[n] + res_scale * res[z][m][n]
 * with A the implicit im2col of x[z] = [B,H,W,Cin] (pixel stride ldx), m = (b,oh,ow),
 * k = (r,s,ci), zero padding `pad`, stride `stride`.  R=S=1,pad=0 gives a plain GEMM.
 * z = zb*nh + zh with element strides (s*b, s*h) per operand.  out_f32: y is fp32 else T.
 * Replaces nn.Conv2d / nn.Conv1d(k=1) forward and -- with the flipped/transposed weight
 * copy of jg_refresh_weights -- its input-gradient
 * (models/modules/unet_generator_attn/unet_generator_attn.py:186-220,297,305,482,635-642),
 * and the two einsum matmuls of QKVAttentionLegacy.forward (same file :342-346). */
typedef struct {
  const void* x; const void* w; void* y; const float* bias; const void* res;
  int32_t B, H, W, Cin, Cout, R, S, pad, stride, Ho, Wo;
  int64_t ldx, ldw, ldy, ldres;
  int32_t nbatch, nh;
  int64_t sxb, sxh, swb, swh, syb, syh, srb, srh;
  float alpha, res_scale;
  int32_t out_f32;
  /* optional fused GroupNorm statistics of the OUTPUT (the consumer's jg_gn_stats pass folded into
   * this epilogue): stats[(b * ldstats + n) * 2 + {0,1}] += (sum, sum of squares) over the pixels of
   * image b, fp32 atomics, computed from the fp32 values before the 16-bit rounding.  The caller
   * zeroes the buffer.  Needs nbatch == 1, out_f32 == 0, Ho*Wo % 256 == 0, Cout % 64 == 0
   * (JG_ERR_UNSUPPORTED otherwise).  ldstats = channel count of the statistics row (>= Cout: lets
   * two producers fill the halves of a channel-concatenated consumer); 0 means Cout. */
  float* stats;
  int64_t ldstats;
  /* stats_slots > 1: the statistics row of image b is replicated stats_slots times
   * (stats[((b * stats_slots + slot) * ldstats + n) * 2 + ...], slot = tile index % stats_slots) so that
   * the same-address atomic chains of a high-resolution layer are stats_slots times shorter; the
   * consumer (jg_gn_coef_ld) sums the slots. */
  int32_t stats_slots;
  /* stats_mode 1: instead of (sum y, sum y^2) the epilogue accumulates the GroupNorm-BACKWARD reductions
   * of the norm whose output-gradient this call computes (y = dL/d act(a*gx+b), i.e. this is the
   * input-gradient convolution of the layer that consumed the norm's output):
   *   stats[..][0] += sum du,  stats[..][1] += sum du * gx,   du = y * act'(a*gx + b)
   * with gn_x the norm's INPUT [B,Ho,Wo,Cout] (pixel stride gn_ldx), gn_ab its (a, b) coefficients
   * [B][Cout][2] from jg_gn_coef, gn_act its activation: the separate jg_gn_bwd_reduce pass over
   * (x, dy) is folded into this epilogue (one extra 16-byte read of x per output row). */
  int32_t stats_mode;
  const void* gn_x; int64_t gn_ldx; const float* gn_ab; int32_t gn_act;
  /* pad_mode 1: out-of-image taps read the MIRRORED interior pixel instead of zero -- nn.ReflectionPad2d(1) followed by a
   * pad-0 3x3 convolution (resnet_generator.py:52-60) as ONE launch (pad = 1, Ho = H, Wo = W).  Halo-resident kernel only:
   * 3x3, stride 1, Cin % 64 == 0, Cout % 64 == 0, H % 16 == 0, W % 16 == 0, else JG_ERR_UNSUPPORTED. */
  int32_t pad_mode;
  /* res_mode 1: res is [B, Ho/2, Wo/2, Cout] and is read through the nearest-upsample map, i.e. y += res_scale * Upsample(res)
   * without the upsampled copy (ResBlock-up skip path `self.x_upd(x)`, unet_generator_attn.py:236-246).  Ho, Wo even, nbatch 1. */
  int32_t res_mode;
  /* x_mode 1: x is [B, H/2, W/2, Cin] and the convolution runs over its nearest x2 upsample (H, W = the upsampled size, even) without
   * the upsampled copy: `conv(Upsample(h))` of the ResBlock-up path (unet_generator_attn.py:120-140,239-246).  Halo-resident 3x3
   * kernel only (shape limits of pad_mode 1), else JG_ERR_UNSUPPORTED.
   * x_mode 2: the same function in its sub-pixel form: w is the FOLDED weight tensor [4][Cout][2][2][Cin] of jg_subpixel_fold
   * (ldw = 4 * Cin), every output phase (oh & 1, ow & 1) is a 2x2-tap convolution of the half-resolution x -- 16 instead of 36
   * tap-MACs per input pixel.  H, W multiples of 32. */
  int32_t x_mode;
  /* y_mode 1: y is [B, Ho/2, Wo/2, Cout] and receives the 2x2 SUM-pool of alpha * conv(x) -- the adjoint of x_mode 1, i.e. the
   * input gradient of `conv(Upsample(h))` w.r.t. h, without the full-resolution gradient in between.  No bias / res / stats;
   * halo-resident 3x3 kernel only, else JG_ERR_UNSUPPORTED. */
  int32_t y_mode;
  /* optional fp32 scratch for the split-K form of the generic kernel (16-byte aligned, ws_bytes long, contents irrelevant on entry and
   * on return).  Layers with few output tiles and a long reduction -- the discriminators' 4x4 stride-2 convolutions at 16x16 .. 4x4
   * (projected_d/discriminator.py DownBlock / SingleDisc, NLayerDiscriminator's last layers) -- leave most of the 256 CUs idle with one
   * workgroup per output tile; with a workspace the launch is cut into K slices whose partial tiles are summed in slice order by a
   * second, element-wise launch: same result on every run.  NULL = never split.  Not combined with stats / res_mode 1. */
  void* ws; int64_t ws_bytes;
} jg_conv_args;
int jg_conv2d_nt(int dtype, const jg_conv_args* a, jg_stream_t stream);
/* 1x1 convolution of x WITH the GroupNorm apply pass of x in the same launch (round 4): y = conv(x) exactly as jg_conv2d_nt, plus
 * y_norm[m][c] = act(ab[b][c][0] * x[m][c] + ab[b][c][1]) for the Cin input channels (pixel stride ldyn, ab from jg_gn_coef).  A ResBlock
 * whose channel count changes reads its input twice -- `self.skip_connection(x)` and `self.in_layers(x)` = conv(SiLU(GroupNorm(x)))
 * (unet_generator_attn.py:233-266) -- here both readers share ONE pass over x.  Only the shapes of the streaming 1x1 kernel (R = S = 1,
 * stride 1, Cin % 32 == 0, Cin <= 256, Cout % 64 == 0, >= 65536 pixels, H W % 16 == 0, no statistics): JG_ERR_UNSUPPORTED otherwise
 * (nothing is launched; callers fall back to jg_gn_apply_ld + jg_conv2d_nt). */
int jg_conv1x1_gn_apply(int dtype, const jg_conv_args* a, const float* ab, void* y_norm, int64_t ldyn, int act, jg_stream_t stream);
/* The backward counterpart: the input gradient of the 1x1 skip convolution WITH the GroupNorm-backward apply step of the same tensor in
 * its epilogue,  y = alpha (dO . W^T) + du P + gx Q + R (+ scale1 add1 + scale2 add2),  du = gdy act'(a gx + b),  (a, b) = ab[b][c] and
 * (P, Q, R) = pqr[b][c] of jg_gn_coef / jg_gn_bwd_coef(_slots): replaces jg_gn_bwd_apply_ld followed by jg_conv2d_nt(res = its result) --
 * the gradient fan-in of a ResBlock input that feeds `in_layers` and `skip_connection` (unet_generator_attn.py:233-266) without the
 * intermediate tensor (one write and one read of [M][Cout] less).  a->x = dO, a->w = the flipped / transposed weights, a->Cout = channel
 * count of gx / gdy / add1 / add2 / y; a->bias and a->res must be NULL.  Streaming-kernel shapes only, else JG_ERR_UNSUPPORTED. */
int jg_conv1x1_gn_bwd_apply(int dtype, const jg_conv_args* a, const void* gn_x, int64_t ldgx, const void* gn_dy, int64_t ldgdy, const float* ab,
                            const float* pqr, const void* add1, int64_t ldadd1, float scale1, const void* add2, int64_t ldadd2, float scale2,
                            int act, jg_stream_t stream);
/* Folded weights of x_mode 2 from the fp32 master weights w32 [Cout][3][3][Cin]: out[py*2+px][co][a][b][ci] (dtype) = sum of the
 * 3x3 taps that land on tap (a, b) of output phase (py, px) of conv3x3(Upsample_nearest(x)) -- per axis {w0 | w1+w2} for phase 0,
 * {w0+w1 | w2} for phase 1 (fp32 sum, one rounding). */
int jg_subpixel_fold(int dtype, const float* w32, void* out, int Cout, int Cin, jg_stream_t s);
/* Four-phase (sub-pixel) form of a stride-2 TRANSPOSED convolution -- nn.ConvTranspose2d(k 3, stride 2, padding 1, output_padding 1) of the
 * ResnetDecoder tail (models/modules/segformer/segformer_generator.py:135-140) and the input gradient of the discriminators' Conv2d(k 4,
 * stride 2, padding 1) (models/modules/discriminators.py NLayerDiscriminator): wT = the flipped / transposed 16-bit working weights
 * [N][R][S][C]; out [4][N][2][2][C] for jg_conv2d_nt with x_mode 2 (R = S = 3, pad 1 in the args; ldw = 4 C).  Replaces the zero-dilated
 * copy of the input + a stride-1 convolution over it. */
int jg_transposed_fold(int dtype, const void* wT, void* out, int N, int C, int R, int S, int pad, jg_stream_t s);

/* Weight gradient / batched GEMM "TN" on MFMA:
 *   dw[z][co][(r,s,ci)] (+)= alpha * sum_p dy[z][p][co] * xcol[z][p][(r,s,ci)]
 * p = (b,oh,ow).  Output fp32 KRSC with row stride lddw and inner channel count Cin_out
 * (<= Cin; lets a channel-padded activation feed an unpadded master gradient), rows
 * co < Cout_out.  out_mode JG_OUT_ATOMIC_F32 accumulates (split-K + gradient accumulation),
 * STORE modes need splitk == 1.  dbias (optional, atomic): dbias[co] += sum_p dy[p][co].
 * Replaces the weight/bias gradient of nn.Conv2d/Conv1d (autograd of the layers above) and
 * the transposed matmuls of the attention backward. */
typedef struct {
  const void* dy; const void* x; void* dw; float* dbias;
  int32_t B, H, W, Cin, Cout, R, S, pad, stride, Ho, Wo;
  int32_t Cin_out, Cout_out;
  int64_t lddy, ldx, lddw;
  int32_t nbatch, nh, splitk;
  int64_t sdyb, sdyh, sxb, sxh, sdwb, sdwh;
  float alpha;
  int32_t out_mode;
  float dbias_scale; /* dbias[co] += dbias_scale * sum_p dy[p][co]; 0 means 1 (a skip-path conv fed with skipw*dy) */
  int32_t pad_mode;  /* 1: x is read with mirrored borders (see jg_conv_args.pad_mode); same shape limits */
  int32_t x_mode;    /* 1: x is the half-resolution tensor of jg_conv_args.x_mode 1 (H, W = the upsampled size); same shape limits */
} jg_wgrad_args;
int jg_conv2d_wgrad_tn(int dtype, const jg_wgrad_args* a, jg_stream_t stream);
/* `n` <= 4096 independent weight-gradient problems (any geometry of jg_conv2d_wgrad_tn with nbatch 1, out_mode JG_OUT_ATOMIC_F32, pad_mode 0,
 * x_mode 0; those the halo-resident 3x3 / 7x7 kernels serve are launched by them, singly) in grouped launches of up to 16: the weight / bias gradients of the SegFormer generator's ~190 linear layers and 1x1 convolutions per cut_model step
 * (models/modules/segformer/backbone.py:13-88,239-328: autograd's per-layer `grad_weight` GEMMs), which singly occupy 8 - 64 workgroups for
 * 10 - 40 us each.  Descriptors travel by value in the kernel argument block (capturable in a hipGraph as they are).  Same arithmetic per
 * problem as jg_conv2d_wgrad_tn; JG_ERR_UNSUPPORTED for any other mode. */
int jg_conv2d_wgrad_tn_group(int dtype, const jg_wgrad_args* a, int n, jg_stream_t stream);

/* GroupNorm (+FiLM scale-shift) (+SiLU), NHWC, statistics in fp32.
 * Replaces GroupNorm.forward (unet_attn_utils.py:42-48), `h*(1+scale)+shift`
 * (unet_generator_attn.py:254-258), torch.nn.SiLU, and nn.InstanceNorm1d (groups == C,
 * gamma = beta = NULL; unet_attn_utils.py:60-66). x: [B, HW, C] T.
 *   stats : sums[B][C][2] = (sum x, sum x^2)                 (zeroed inside)
 *   coef  : ab[B][C][2]: y = act(a*x + b); mr[B][G][2] = (mean, rstd)
 *           film = emb_out[B][2C] (scale | shift) with row stride ldfilm, or NULL
 *   apply : y = act(a*x+b) */
int jg_gn_stats(int dtype, const void* x, float* sums, int B, int HW, int C, jg_stream_t s);
int jg_gn_coef(const float* sums, const float* gamma, const float* beta, const float* film, int64_t ldfilm,
               float* ab, float* mr, int B, int HW, int C, int G, float eps, jg_stream_t s);
/* Round 6: jg_gn_apply_ld with a residual addend: y = act(a x + b) + add (add [B, HW, C], pixel stride ldadd) -- `x + self.conv_block(x)` of
 * ResnetBlock / resnet_block_attn (resnet_generator.py:11-95, 350-385), whose branch ends in an InstanceNorm, in the norm's apply pass. */
int jg_gn_apply_add(int dtype, const void* x, int64_t ldx, const float* ab, const void* add, int64_t ldadd, void* y, int64_t ldy, int B, int HW,
                    int C, int act, jg_stream_t s);
int jg_gn_apply(int dtype, const void* x, const float* ab, void* y, int B, int HW, int C, int act, jg_stream_t s);
/* backward: red[B][C][2] = (sum du, sum du*x), du = dy * act'(a*x+b)      (zeroed inside)
 *   bwd_coef: pqr[B][C][3] so that dx = du*P + x*Q + R;  dgamma/dbeta += (atomic, may be NULL);
 *             dfilm[B][2C] (row stride lddfilm) = (d scale | d shift) written (not accumulated)
 *   bwd_apply: dx */
int jg_gn_bwd_reduce(int dtype, const void* x, const void* dy, const float* ab, float* red,
                     int B, int HW, int C, int act, jg_stream_t s);
int jg_gn_bwd_coef(const float* red, const float* gamma, const float* beta, const float* film, int64_t ldfilm,
                   const float* mr, float* pqr, float* dgamma, float* dbeta, float* dfilm, int64_t lddfilm,
                   int B, int HW, int C, int G, jg_stream_t s);
int jg_gn_bwd_apply(int dtype, const void* x, const void* dy, const float* ab, const float* pqr, void* dx,
                    int B, int HW, int C, int act, jg_stream_t s);
/* bwd_coef on reductions that were accumulated by a convolution epilogue (jg_conv_args.stats_mode 1):
 * red rows are replicated nslots times, [(b*nslots + slot)*C + c][2], and summed here. */
int jg_gn_bwd_coef_slots(const float* red, int nslots, const float* gamma, const float* beta, const float* film,
                         int64_t ldfilm, const float* mr, float* pqr, float* dgamma, float* dbeta, float* dfilm,
                         int64_t lddfilm, int B, int HW, int C, int G, jg_stream_t s);
/* Strided forms of the three streaming passes: every tensor carries its own pixel stride (elements,
 * multiple of 8, >= C), so they run on channel slices of a concatenated buffer without a copy
 * (torch.cat of the UNet skip connections, unet_generator_attn.py:692-693, and its backward split).
 * bwd_apply_ld additionally folds the gradient accumulation of the OTHER consumers of x into the
 * same pass: dx = GN-backward(...) + scale1*add1 + scale2*add2 (each optional) -- the residual /
 * skip-connection / concat fan-out sums that autograd would run as separate add kernels. */
/* stats_ld ACCUMULATES into sums (the caller zeroes), x pixel stride ldx, statistics row stride ldsums channels. */
int jg_gn_stats_ld(int dtype, const void* x, int64_t ldx, float* sums, int64_t ldsums, int B, int HW, int C, jg_stream_t s);
/* coef_ld: the statistics rows of image b start at sums + b*nslots*ldsums*2 (nslots replicas that are summed; channel slice of a wider row);
 * HW is the pixel count the sums were taken over (a nearest-upsampled tensor reuses the sums of
 * its source with the source's HW: mean and variance are unchanged). */
int jg_gn_coef_ld(const float* sums, int64_t ldsums, int nslots, const float* gamma, const float* beta, const float* film,
                  int64_t ldfilm, float* ab, float* mr, int B, int HW, int C, int G, float eps, jg_stream_t s);
int jg_gn_apply_ld(int dtype, const void* x, int64_t ldx, const float* ab, void* y, int64_t ldy,
                   int B, int HW, int C, int act, jg_stream_t s);
int jg_gn_bwd_reduce_ld(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* ab,
                        float* red, int B, int HW, int C, int act, jg_stream_t s);
int jg_gn_bwd_apply_ld(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* ab,
                       const float* pqr, void* dx, int64_t lddx, const void* add1, int64_t ldadd1, float scale1,
                       const void* add2, int64_t ldadd2, float scale2, int B, int HW, int C, int act, jg_stream_t s);

/* y[B, H/2, W/2, C] = scale * sum_{2x2} act(a * x + b): GroupNorm apply + activation + 2x2 pool in one pass (ResBlock-down `h` path,
 * unet_generator_attn.py:239-246 with Downsample = AvgPool2d at :163-176). */
int jg_gn_apply_pool(int dtype, const void* x, int64_t ldx, const float* ab, void* y, int64_t ldy, int B, int H, int W, int C, int act,
                     float scale, jg_stream_t s);

/* GroupNorm backward of a layer whose output went through nn.AvgPool2d(2,2) (the ResBlock-down path `pool(act(norm(x)))`,
 * unet_generator_attn.py:239-246): dy_low / add1_low are gradients at the POOLED resolution [B, H/2, W/2, C]; the pool's adjoint
 * (nearest upsample * dy_scale, .25 for the average) is applied while reading, so the full-resolution gradient is never written.
 * add2 (optional) is a full-resolution addend as in jg_gn_bwd_apply_ld. */
int jg_gn_bwd_reduce_up(int dtype, const void* x, int64_t ldx, const void* dy_low, int64_t lddy, float dy_scale, const float* ab,
                        float* red, int B, int H, int W, int C, int act, jg_stream_t s);
/* jg_gn_bwd_apply_ld / _up with the coefficient step of jg_gn_bwd_coef inside (one launch fewer per GroupNorm backward): (P, Q, R) are derived per
 * workgroup from `red` ([B][C][2], one slot), gamma / beta / FiLM and mr; dgamma / dbeta / dfilm come out as from jg_gn_bwd_coef. */
int jg_gn_bwd_apply_fc(int dtype, int up, const void* x, int64_t ldx, const void* dy, int64_t lddy, float dy_scale, const float* ab, const float* red,
                       const float* gamma, const float* beta, const float* film, int64_t ldfilm, const float* mr, float* dgamma, float* dbeta,
                       float* dfilm, int64_t lddfilm, int G, void* dx, int64_t lddx, const void* add1, int64_t ldadd1, float scale1,
                       const void* add2, int64_t ldadd2, float scale2, int B, int H, int W, int C, int act, jg_stream_t s);
/* GroupNorm backward in ONE launch with x and dy resident on chip between the reduction and the apply step (csrc/gn_fused.hip): replaces
 * jg_gn_bwd_reduce_* + jg_gn_bwd_coef + jg_gn_bwd_apply_* (autograd of `GroupNorm32` + FiLM + SiLU, unet_attn_utils.py:42-48,
 * unet_generator_attn.py:233-266) at 6 instead of 10 bytes per element.  `red` ([B][C][2] floats) and `counters` ([B] words: arrival
 * counter of every image's workgroups) must be ZERO at launch; *status (one word, sticky) becomes 1 if an inter-workgroup wait expired
 * (results of that launch are then invalid; the kernel never hangs).  up != 0: dy / add1 at the 2x2-pooled resolution. */
int jg_gn_bwd_fused(int dtype, int up, const void* x, int64_t ldx, const void* dy, int64_t lddy, float dy_scale, const float* ab, float* red,
                    uint32_t* counters, uint32_t* status, const float* gamma, const float* beta, const float* film, int64_t ldfilm,
                    const float* mr, float* dgamma, float* dbeta, float* dfilm, int64_t lddfilm, int G, void* dx, int64_t lddx,
                    const void* add1, int64_t ldadd1, float scale1, const void* add2, int64_t ldadd2, float scale2, int B, int H, int W,
                    int C, int act, jg_stream_t s);
/* jg_gn_bwd_reduce_ld / _up ACCUMULATING into `red` instead of clearing it first: the UNet executor hands over zeroed rows of a pool it
 * clears once per backward pass (57 memset launches per step fewer) */
int jg_gn_bwd_reduce_ld_acc(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, const float* ab, float* red, int B, int HW,
                            int C, int act, jg_stream_t s);
int jg_gn_bwd_reduce_up_acc(int dtype, const void* x, int64_t ldx, const void* dy_low, int64_t lddy, float dy_scale, const float* ab,
                            float* red, int B, int H, int W, int C, int act, jg_stream_t s);
int jg_gn_bwd_apply_up(int dtype, const void* x, int64_t ldx, const void* dy_low, int64_t lddy, float dy_scale, const float* ab,
                       const float* pqr, void* dx, int64_t lddx, const void* add1_low, int64_t ldadd1, float scale1, const void* add2,
                       int64_t ldadd2, float scale2, int B, int H, int W, int C, int act, jg_stream_t s);

/* 2x2 sum-pool * scale and nearest x2 upsample * scale, NHWC.  Forward/backward of
 * nn.AvgPool2d(2,2) (pool scale .25 / upsample scale .25) and of
 * F.interpolate(scale_factor=2, mode="nearest") (upsample scale 1 / pool scale 1)
 * (unet_generator_attn.py:81-96,121-140). */
int jg_pool2x2(int dtype, const void* x, void* y, int B, int H, int W, int C, float scale, jg_stream_t s);
int jg_upsample2x(int dtype, const void* x, void* y, int B, int H, int W, int C, float scale, jg_stream_t s);
/* the same with explicit pixel strides (channel slices of a concatenated buffer) */
int jg_pool2x2_ld(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C, float scale,
                  jg_stream_t s);
int jg_upsample2x_ld(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int B, int H, int W, int C, float scale,
                     jg_stream_t s);
/* dst[p][doff:doff+n] = src[p][soff:soff+n] for p < P (channel-block copy; torch.cat(dim=1) and its
 * backward split in NHWC; unet_generator_attn.py:687). */
int jg_copy_channels(int dtype, const void* src, int64_t ldsrc, int64_t soff, void* dst, int64_t lddst, int64_t doff,
                     int64_t P, int n, jg_stream_t s);
/* y = alpha * (*alpha_dev) * a + beta * b over n elements (b and alpha_dev may be NULL): gradient
 * accumulation at a fork, the 1/sqrt(2) skip scale of the "efficient" ResBlock
 * (unet_generator_attn.py:263-266) and the chain-rule scale of the loss gradient by a device scalar. */
int jg_axpby(int dtype, const void* a, float alpha, const float* alpha_dev, const void* b, float beta, void* y,
             int64_t n, jg_stream_t s);

/* Attention helpers (QKVAttentionLegacy.forward, unet_generator_attn.py:331-347).
 *   transpose_heads: dst[(b*nh+h)][c][t] = src[b][t][coff + h*hstride + c], c < ch
 *   softmax_fwd: P[row][:] = softmax(S[row][:]) (S fp32 -> P T), rows x T
 *   softmax_bwd: dS = alpha * P * (dP - sum_j dP_j P_j)  (dP fp32 -> dS T) */
int jg_transpose_heads(int dtype, const void* src, int64_t ldsrc, int64_t coff, int64_t hstride, void* dst,
                       int B, int T, int nh, int ch, jg_stream_t s);
int jg_softmax_fwd(int dtype, const float* S, void* P, int64_t rows, int T, jg_stream_t s);
int jg_softmax_bwd(int dtype, const void* P, const float* dP, void* dS, int64_t rows, int T, float alpha, jg_stream_t s);

/* Fused self-attention (flash style, head_dim 32, T % 128 == 0; JG_ERR_UNSUPPORTED otherwise -> use the
 * jg_conv2d_nt + jg_softmax pipeline above).  qkv [B,T,3C] in the legacy layout of QKVAttentionLegacy
 * (channel = h*3*ch + {q,k,v}*ch + c, unet_generator_attn.py:340); a [B,T,C] (channel = h*ch + c);
 * L [B*heads, T] fp32 = logsumexp of the scaled logits (kept for the backward instead of the T x T matrix).
 * Replaces QKVAttentionLegacy.forward (:331-347) and its autograd: the logits never reach HBM.
 * bwd: o = the forward output, da its gradient; dqkv fully written; Dq [B*heads, T] fp32 scratch. */
int jg_attention_fwd(int dtype, const void* qkv, void* a, float* L, int B, int T, int heads, int head_dim, jg_stream_t s);
int jg_attention_bwd(int dtype, const void* qkv, const void* o, const float* L, const void* da, void* dqkv, float* Dq,
                     int B, int T, int heads, int head_dim, jg_stream_t s);

/* Small fp32 linear layers on the embedding path (nn.Linear in ResBlock.emb_layers and
 * DiffusionGenerator.cond_embed; unet_generator_attn.py:201-207, diffusion_generator.py:74-78).
 *   fwd: y[b][n] = sum_k act(x[b][k]) W[n][k] + bias[n]
 *   bwd: dx[b][k] = act'(x[b][k]) sum_n dy[b][n] W[n][k] (dx may be NULL);
 *        dW[n][k] += sum_b dy[b][n] act(x[b][k]); dbias[n] += sum_b dy[b][n]  (may be NULL) */
int jg_linear_fwd(const float* x, const float* W, const float* bias, float* y, int Bn, int K, int N, int act, jg_stream_t s);
int jg_linear_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* dbias,
                  int Bn, int K, int N, int act, jg_stream_t s);
/* gamma_embedding (models/modules/diffusion_utils.py:8-42): emb[b] = cat(cos(g f), sin(g f)). */
int jg_gamma_embedding(const float* gammas, float* emb, int Bn, int dim, float max_period, jg_stream_t s);

/* DDPM glue (DiffusionGenerator.forward, models/modules/diffusion_generator.py:467-491):
 * y_noisy = sqrt(g) y0 + sqrt(1-g) noise; blended with clamp(mask,0,1); concatenated after
 * y_cond and written NHWC with Cpad (=8) channels, zero padded.  y0/y_cond/noise: NCHW fp32
 * [B,3,H,W]; mask int64 [B,1,H,W] or NULL; gammas [B]. */
int jg_ddpm_prepare(int dtype, const float* y0, const float* ycond, const float* noise, const int64_t* mask,
                    const float* gammas, void* xin, int B, int C, int H, int W, int Cpad, jg_stream_t s);
/* Masked / min-SNR-weighted MSE of PaletteModel.compute_palette_loss (models/palette_model.py:597-620):
 * loss += lambda * mean((w m noise - w m noise_hat)^2) (atomic into *loss, caller zeroes);
 * dnh (NHWC Cpad, T) = grad_scale * d loss / d noise_hat.  w [B] or NULL. */
int jg_ddpm_mse_loss(int dtype, const float* noise, const void* noise_hat, const int64_t* mask, const float* w,
                     float* loss, void* dnh, int B, int C, int H, int W, int Cpad, float lambda, float grad_scale,
                     jg_stream_t s);
/* alg_palette_loss in {L1, multiscale_L1, multiscale_MSE} (palette_model.py:231-256,597-618; MultiScaleDiffusionLoss
 * models/modules/loss.py:397-467) on d = w m (noise - noise_hat): level l (factor f = 2^l, l < nlevels) is the loss of the
 * bilinear down-sampling of d to S / f -- for integer factors the mean of the central 2x2 pixels of each f x f cell --
 * weighted 32 / (2 S / f); multiscale = 0: a single full-resolution level with weight 1 (plain nn.L1Loss / nn.MSELoss).
 * losses[nlevels] (zeroed by the caller) receive lambda * the per-level terms (their sum is loss_G_tot); dnh (may be NULL) =
 * grad_scale * d(sum)/d noise_hat.  ws: caller-provided fp32 workspace of sum_{l>=1} B C (H >> l)(W >> l) elements. */
int jg_ddpm_multiscale_loss(int dtype, const float* noise, const void* noise_hat, const int64_t* mask, const float* w,
                            float* losses, void* dnh, float* ws, int B, int C, int H, int W, int Cpad, int nlevels, int l1,
                            int multiscale, float lambda, float grad_scale, jg_stream_t s);
/* Glue of the CUT networks (ResnetGenerator models/modules/resnet_architecture/resnet_generator.py:11-347,
 * NLayerDiscriminator models/modules/discriminators.py:10-118), NHWC 16-bit, C % 8 == 0:
 *   act_fwd / act_bwd        : nn.ReLU / nn.LeakyReLU(0.2) / nn.Tanh where no normalisation precedes them; the backward
 *                              takes the OUTPUT y (tanh' = 1 - y^2)
 *   reflect_pad2d (+ _bwd)   : nn.ReflectionPad2d(pad) and its adjoint (gather form, deterministic)
 *   dilate2d / subsample2d   : zero insertion y[s*i, s*j] = x[i, j] into [Ho, Wo] and its adjoint: a stride-s
 *                              nn.ConvTranspose2d (and the input gradient of a stride-s nn.Conv2d) is jg_conv2d_nt at
 *                              stride 1 over the dilated tensor with the flipped / transposed weight copy
 *   channel_sum              : out[c] += scale * sum_p x[p][c] (bias gradient of the transposed convolution) */
int jg_act_fwd(int dtype, const void* x, void* y, int64_t n, int act, jg_stream_t s);
int jg_act_bwd(int dtype, const void* y, const void* dy, void* dx, int64_t n, int act, jg_stream_t s);
int jg_reflect_pad2d(int dtype, const void* x, void* y, int B, int H, int W, int C, int pad, jg_stream_t s);
int jg_reflect_pad2d_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int pad, jg_stream_t s);
/* Input gradient of ReflectionPad2d(1) -> Conv2d(3x3, padding 0) (the ResnetBlocks of the CUT generators,
 * models/modules/resnet_architecture/resnet_generator.py:247-275; autograd of F.pad(mode="reflect") + F.conv2d in the reference):
 * dx holds the zero-padded ("same") input gradient on the H x W domain (jg_conv2d_nt on the flipped / transposed weight copy wT
 * [Cin][3][3][Cout] with pad 1 -- the halo-resident kernel's shape); this adds alpha x the one-pixel ring of the padded-domain gradient,
 * folded by the reflection's adjoint, into rows 1 / H - 2 and columns 1 / W - 2 of dx (csrc/reflect_border.hip).  Together: what
 * jg_conv2d_nt over the (H + 2) x (W + 2) domain followed by jg_reflect_pad2d_bwd computes.  Cout % 32 == 0, Cin % 8 == 0,
 * H, W <= 272; no atomics (every target pixel is increased once, by one thread). */
int jg_reflect_dgrad_border(int dtype, const void* dy, int64_t lddy, const void* wT, void* dx, int64_t lddx, float* ws, int B, int H, int W,
                            int Cout, int Cin, float alpha, jg_stream_t s);
/* fp32 scratch of jg_reflect_dgrad_border ([B][4 lines][max(H, W) + 2][Cin]) */
int64_t jg_reflect_dgrad_border_ws_floats(int B, int H, int W, int Cin);
/* window crop (adjoint = 0: y[B,Ho,Wo,C] = x[B,H,W,C][:, top:top+Ho, left:left+Wo]) and its adjoint (1: y[B,H,W,C] = x[B,Ho,Wo,C] placed
 * at (top, left), zeros elsewhere): reflect-padded depth-wise conv of SeparableConv2d (mobile_modules.py:4-40) = pad -> dwconv -> crop */
int jg_crop2d(int dtype, const void* x, void* y, int B, int H, int W, int C, int top, int left, int Ho, int Wo, int adjoint, jg_stream_t s);
/* Round 6, row-packed 7x7 head: nn.ReflectionPad2d(3) + nn.Conv2d(64, 3, 7) (+ nn.Tanh), the last layer of the CUT generators
 * (resnet_generator.py:247-263, segformer_generator.py:135-140), as a 1 x 7 convolution onto 7 x 4 packed channels (jg_conv2d_nt, R = 1, S = 7:
 * z[B][H+6][W][32], column ky * 4 + c = tap row ky of output channel c) followed by the sum over the tap rows:
 *   jg_tapsum7:    out[b][y][x][c] = act(bias[c] + sum_ky z[b][y + ky][x][ky * 4 + c]),  out [B][H][W][8], act JG_ACT_NONE / JG_ACT_TANH
 *   jg_tapspread7: its adjoint -- dz[b][y'][x][ky * 4 + c] = dout[b][y' - ky][x][c] * act'(out), written as dz [B][Hz][W][32] (Hz >= H + 6, a
 *                  multiple of 8: operand of jg_conv2d_wgrad_tn with R = 1, S = 7) and as dzm [B][H+6][W+12][32] with six zero columns on
 *                  either side (operand of the 1 x 7 input-gradient convolution, pad 0). */
int jg_tapsum7(int dtype, const void* z, const float* bias, void* out, int B, int H, int W, int act, jg_stream_t s);
int jg_tapspread7(int dtype, const void* dout, const void* out, void* dz, void* dzm, int B, int H, int W, int Hz, int act, jg_stream_t s);
int jg_dilate2d(int dtype, const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int stride, jg_stream_t s);
/* Round 6: autograd's grad_input of a STRIDED nn.Conv2d whose input is an image (<= 4 real channels in an 8-channel NHWC pixel) -- the first
 * convolution of the MiT encoder (segformer/backbone.py PatchEmbed: 7x7 stride 4) and of NLayerDiscriminator (discriminators.py:53-60: 4x4
 * stride 2) -- in gather form: every input pixel sums the ceil(R/s) x ceil(S/s) taps that reach it.  dy [B,Ho,Wo,Cout], w16 [Cout][R][S][8]
 * (forward layout), dx [B,H,W,8] (channels 4..7 written as zero).  JG_ERR_UNSUPPORTED when R*S*Cout*16 bytes exceed 64 KB of LDS. */
int jg_conv_dgrad_gather(int dtype, const void* dy, const void* w16, void* dx, int B, int H, int W, int Ho, int Wo, int Cout, int R, int S,
                         int stride, int pad, float alpha, jg_stream_t s);

/* Frozen tf_efficientnet_lite0 feature network of the projected discriminator (models/modules/projected_d/projector.py:51-59,251-255; timm
 * MBConv blocks; the feature network stays in eval mode, discriminator.py:267-270, so BatchNorm is the per-channel affine scale / shift):
 *   dwconv_affine_act : y = act(scale[c] * dwconv_kxk(x, w)[c] + shift[c]); k = 3 | 5, stride 1 | 2, explicit top / left zero padding
 *                       (TF "SAME" is asymmetric at stride 2), w = fp32 [k*k][C], act 0 = none, 1 = ReLU6; _bwd = the INPUT gradient
 *                       (weights are frozen) from dy and the forward output y (ReLU6 pass-through set = 0 < y < 6)
 *   chan_affine_act   : y = act(scale[c] * x + shift[c]) over P pixels (BatchNorm + ReLU6 behind the 1x1 convolutions) and its gradient */
int jg_dwconv_affine_act_fwd(int dtype, const void* x, const float* w, const float* scale, const float* shift, void* y, int B, int H, int W, int C,
                             int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, jg_stream_t s);
int jg_dwconv_affine_act_bwd(int dtype, const void* dy, const void* y, const float* w, const float* scale, void* dx, int B, int H, int W, int C,
                             int k, int stride, int pad_t, int pad_l, int Ho, int Wo, int act, jg_stream_t s);
int jg_chan_affine_act_fwd(int dtype, const void* x, const float* scale, const float* shift, void* y, int64_t P, int C, int act, jg_stream_t s);
int jg_chan_affine_act_bwd(int dtype, const void* dy, const void* y, const float* scale, void* dx, int64_t P, int C, int act, jg_stream_t s);
int jg_subsample2d(int dtype, const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int stride, jg_stream_t s);
int jg_channel_sum(int dtype, const void* x, int64_t ldx, float* out, int64_t P, int C, float scale, jg_stream_t s);

/* PatchSampleF (models/modules/cut_networks.py:6-73) and GANLoss "lsgan" (models/modules/loss.py:59-85) glue:
 *   gather_rows / scatter_rows : feat[B,HW,C] -> [B*P, C] fp32 at the shared patch ids (:43-57) and its adjoint (ids unique)
 *   l2norm_fwd / bwd           : torch.nn.functional.normalize(x, eps=1e-7) over the last dim (:66), fp32
 *   jg_linear_* with act = JG_ACT_RELU : the Linear-ReLU-Linear MLP (:26-29)
 *   lsgan_loss                 : loss += scale * mean((pred[..., 0] - target)^2), dpred in the same pass (padding channels 0) */
int jg_gather_rows(int dtype, const void* src, int64_t ld, const int64_t* ids, float* dst, int B, int64_t HW, int C, int P,
                   jg_stream_t s);
int jg_scatter_rows(int dtype, void* dsrc, int64_t ld, const int64_t* ids, const float* ddst, int B, int64_t HW, int C, int P,
                    jg_stream_t s);
/* the same with G id sets ([G][P]): image b uses set (b / per) % G -- one launch for the concatenated batch of both contrastive terms of
 * cut_model.py:708-909 ([translated | identity | source | target] x B images: G = 2, per = B) */
int jg_gather_rows_grouped(int dtype, const void* src, int64_t ld, const int64_t* ids, float* dst, int B, int64_t HW, int C, int P, int G, int per,
                           jg_stream_t s);
int jg_scatter_rows_grouped(int dtype, void* dsrc, int64_t ld, const int64_t* ids, const float* ddst, int B, int64_t HW, int C, int P, int G,
                            int per, jg_stream_t s);
int jg_l2norm_fwd(const float* x, float* y, float* nrm, int64_t R, int D, float eps, jg_stream_t s);
int jg_l2norm_bwd(const float* y, const float* nrm, const float* dy, float* dx, int64_t R, int D, float eps, jg_stream_t s);
int jg_lsgan_loss(int dtype, const void* pred, float target, float* loss, void* dpred, int64_t Npix, int Cpad, float scale,
                  float grad_scale, jg_stream_t s);
/* GANLoss (loss.py:59-76) with mode 0 "lsgan" (= jg_lsgan_loss), 1 "vanilla" (nn.BCEWithLogitsLoss against the label `target`), 2 "wgangp"
 * (-mean for real / +mean for fake; the reference never adds the gradient penalty) */
int jg_gan_loss(int dtype, int mode, const void* pred, float target, float* loss, void* dpred, int64_t Npix, int Cpad, float scale,
                float grad_scale, jg_stream_t s);

/* Strided, batched fp32 GEMM: C[z][m][n] = alpha sum_k actA(A[z][m][k]) actB(B[z][n][k]) + bias[n], then
 * C *= act'(E[z][m][n]) (E shares C's strides), then C += beta C_old.  Element strides s??, batch strides b?.
 * Replaces torch.bmm / nn.Linear on the patch features (cut_networks.py:26-29,61; base_NCE.py:55,64). */
int jg_sgemm(const float* A, const float* B, float* C, const float* bias, const float* E, int M, int N, int K, int64_t sam,
             int64_t sak, int64_t sbn, int64_t sbk, int64_t scm, int64_t scn, int nbatch, int64_t ba, int64_t bb, int64_t bc,
             float alpha, float beta, int act_a, int act_b, int act_e, jg_stream_t s);

/* PatchNCE / MoNCE loss (models/modules/NCE/base_NCE.py:17-77, monce.py:16-33, sinkhorn.py:6-58) on the similarity
 * matrices S[nimg][P][P] = q k^T (k detached):
 *   nce_sinkhorn_fwd : K = exp(S / eps) with the diagonal at exp(-10 / eps); niter Sinkhorn iterations (u = 1/(K v),
 *                      v = 1/(K^T u)); writes K and the histories u_hist[nimg][niter][P], v_hist[nimg][niter+1][P]
 *   nce_ce           : per patch CE([S_ii | S_ij (+ T log(u_i K_ij v_j pm1 + 1e-8)), diagonal -10] / T, target 0) -> loss_rows;
 *                      if dS: dS_ij = grow_i dloss_i/dl_neg_ij (diagonal 0), gpos_i = grow_i dloss_i/dl_pos_i and, for MoNCE
 *                      (u != NULL), gW = dL/d(u_i K_ij v_j).  l_pos sees k detached, l_neg does not (base_NCE.py:52-66).
 *   nce_sinkhorn_bwd : reverse sweep through the iterations; dS += the transport-plan path of the gradient
 *                      (gW is consumed/overwritten; ds_hist, dr_hist [nimg][niter][P] are scratch) */
/* y[r][:] += g[r] x[r][:]  (the l_pos path of dq: dq_i += gpos_i k_i) */
int jg_row_axpy(float* y, const float* g, const float* x, int64_t R, int D, jg_stream_t s);
int jg_nce_sinkhorn_fwd(const float* S, float* K, float* u_hist, float* v_hist, int nimg, int P, int niter, float eps,
                        jg_stream_t s);
int jg_nce_ce(const float* S, const float* u, int64_t ustride, const float* v, int64_t vstride, float* loss_rows, float* dS,
              float* gW, int nimg, int P, float T, float pm1, const float* grow, float* gpos, float eps, jg_stream_t s);
int jg_nce_sinkhorn_bwd(const float* K, const float* u_hist, const float* v_hist, float* gW, float* ds_hist, float* dr_hist,
                        float* dS, int nimg, int P, int niter, jg_stream_t s);

/* ---- SegFormer attention generator (G_netG = segformer_attn_conv; models/modules/segformer/, attn_network.py) -------------------
 * Token sequences [B, N, C] are the NHWC maps [B, H, W, C]; C % 8 == 0.
 *   layernorm_fwd/bwd : nn.LayerNorm(C, eps) of backbone.py:382,402,503,601 (mr[R][2] = mean, rstd saved for the backward;
 *                       dgamma / dbeta accumulate)
 *   dwconv3x3_fwd/bwd : MixFFN's depth-wise positional conv + the nn.GELU() behind it (backbone.py:59-77); `pre` keeps the
 *                       pre-activation; bwd writes du = dy gelu'(pre), dx, and accumulates dw[C][9], dbias
 *   attn_smallkv_*    : the softmax(q k^T / sqrt(d)) v core of nn.MultiheadAttention as used by EfficientMultiheadAttention
 *                       (backbone.py:293-313): head dim 32, T_kv <= 256 spatially-reduced keys / values resident in LDS;
 *                       q / o rows of stride ldq / ldo, k and v rows of stride ldkv (views into a packed projection);
 *                       bwd needs fp32 scratch dkf / dvf [B][Tkv][heads*32] and writes 16-bit dk / dv of that shape
 *   bilinear_fwd/bwd  : F.interpolate(mode="bilinear", align_corners=False) of SegformerHead.forward (segformer_head.py:166-174),
 *                       written into a channel slice (row stride ldy) of the concat buffer; bwd for up-sampling
 *   bn_coef/bn_bwd_coef: nn.BatchNorm2d of the ResnetDecoder tail (default norm_layer, segformer_generator.py:135-140) on top of
 *                       the per-(image, channel) sums of jg_gn_stats / jg_gn_bwd_reduce: they fill the ab / pqr coefficient
 *                       tables that jg_gn_apply / jg_gn_bwd_apply consume; training also updates the running statistics
 *   attn_compose_*    : BaseGenerator_attn.forward (attn_network.py:14-46): softmax over the `na` attention logits of every f x f
 *                       cell, out = sum_{i < ni} img[nc i .. nc i + nc) a_i + xin sum_{i >= ni} a_i
 *   jg_scale          : y = x * s (+ res): DropPath (s[B], backbone.py:700-712) / Dropout2d (s[B][C], decode_head.py:181-186) */
int jg_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* mr, int64_t R, int C, float eps,
                     jg_stream_t s);
int jg_layernorm_bwd(int dtype, const void* x, const void* dy, const float* gamma, const float* mr, void* dx, float* dgamma, float* dbeta,
                     int64_t R, int C, jg_stream_t s);
/* the same with the residual branch's gradient added in the pass: dx = res + LayerNorm'(dy) -- the fan-in of a pre-norm block's input,
 * x + drop_path(f(norm(x))) (models/modules/segformer/backbone.py:401-438); res may be NULL */
/* Round 6: nn.LayerNorm of `identity + drop_path(branch)` -- the residual sum in front of every LayerNorm of a pre-norm MiT block
 * (segformer/backbone.py:401-438; DropPath :700-726) -- in one pass: xsum = x + add * scale[row / rows_per_image] (scale NULL: 1), y = LN(xsum)
 * of the ROUNDED sum, mr = (mean, rstd) per row.  Replaces jg_scale (with a residual) + jg_layernorm_fwd. */
int jg_layernorm_fwd_add(int dtype, const void* x, const void* add, const float* scale, int64_t rows_per_image, void* xsum, const float* gamma,
                         const float* beta, void* y, float* mr, int64_t R, int C, float eps, jg_stream_t s);
/* jg_layernorm_bwd_add with a second output dx2 = dx * scale[row / rows_per_image] (scale NULL: a copy): the gradient of the DropPath-scaled
 * branch of the residual sum jg_layernorm_fwd_add formed (round 6; replaces the jg_scale launch over dx). */
int jg_layernorm_bwd_add2(int dtype, const void* x, const void* dy, const float* gamma, const float* mr, const void* res, void* dx,
                          const float* scale, int64_t rows_per_image, void* dx2, float* dgamma, float* dbeta, int64_t R, int C, jg_stream_t s);
int jg_layernorm_bwd_add(int dtype, const void* x, const void* dy, const float* gamma, const float* mr, const void* res, void* dx, float* dgamma,
                         float* dbeta, int64_t R, int C, jg_stream_t s);
int jg_dwconv3x3_fwd(int dtype, const void* x, const float* w, const float* bias, void* pre, void* y, int B, int H, int W, int C, int gelu,
                     jg_stream_t s);
int jg_dwconv3x3_bwd(int dtype, const void* x, const void* pre, const void* dy, const float* w, void* du, void* dx, float* dw, float* dbias,
                     int B, int H, int W, int C, int gelu, jg_stream_t s);
/* the same with the weight / bias partials of the blocks staged in `ws` (>= jg_dwconv3x3_bwd_ws_floats(B, H, W, C) floats, need not be
 * zeroed) and summed by a second launch instead of one atomic per block and destination; ws NULL = jg_dwconv3x3_bwd */
int jg_dwconv3x3_bwd_ws(int dtype, const void* x, const void* pre, const void* dy, const float* w, void* du, void* dx, float* dw,
                        float* dbias, float* ws, int64_t ws_floats, int B, int H, int W, int C, int gelu, jg_stream_t s);
int64_t jg_dwconv3x3_bwd_ws_floats(int B, int H, int W, int C);
/* pad_mode 1: padding_mode = 'reflect' (nn.Conv2d(C, C, 3, padding=1, padding_mode='reflect', groups=C): the depth-wise convolution of
 * SeparableConv2d in the mobile ResNet blocks, models/modules/mobile_modules.py:4-40) -- out-of-image taps read the mirrored pixel, the
 * input gradient collects the mirror images' terms; pad_mode 0 = the entry points above (zero padding). */
int jg_dwconv3x3_fwd_pad(int dtype, const void* x, const float* w, const float* bias, void* pre, void* y, int B, int H, int W, int C, int gelu,
                         int pad_mode, jg_stream_t s);
int jg_dwconv3x3_bwd_ws_pad(int dtype, const void* x, const void* pre, const void* dy, const float* w, void* du, void* dx, float* dw,
                            float* dbias, float* ws, int64_t ws_floats, int B, int H, int W, int C, int gelu, int pad_mode, jg_stream_t s);
int jg_attn_smallkv_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int Tq, int Tkv, int heads,
                        int64_t ldq, int64_t ldkv, int64_t ldo, float scale, jg_stream_t s);
int jg_attn_smallkv_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                        float* dkf, float* dvf, void* dk, void* dv, int B, int Tq, int Tkv, int heads, int64_t ldq, int64_t ldkv, int64_t ldo,
                        float scale, jg_stream_t s);
/* the same with dk / dv written as 16-bit rows of pitch lddkv (dk = buffer, dv = buffer + C of a [B, Tkv, 2 C] gradient of the packed kv
 * projection: no concatenation afterwards); a contiguous (dkf, dvf) accumulator pair is cleared with one fill */
int jg_attn_smallkv_bwd2(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                         float* dkf, float* dvf, void* dk, void* dv, int64_t lddkv, int B, int Tq, int Tkv, int heads, int64_t ldq, int64_t ldkv,
                         int64_t ldo, float scale, jg_stream_t s);
/* the same with align_corners selectable (1: FeatureFusionBlockMatrix of the projected discriminator, projected_d/blocks.py:276-287) */
int jg_bilinear2_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int64_t ldy, int align_corners, jg_stream_t s);
int jg_bilinear2_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo, int64_t lddy, int align_corners, jg_stream_t s);
int jg_bilinear_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int64_t ldy, jg_stream_t s);
int jg_bilinear_bwd(int dtype, const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo, int64_t lddy, jg_stream_t s);
/* Round 6: y[B,Ho,Wo,C] = act(x0 + sum_i F.interpolate(x_i, (Ho, Wo), "bilinear", align_corners=False)) for up to three lower-resolution maps
 * x_i [B,H_i,W_i,C] (NULL = absent); act JG_ACT_NONE / JG_ACT_RELU.  SegformerHead.forward (mmseg decode head used by
 * models/modules/segformer/segformer_generator.py): `fusion_conv(cat([resize(conv_i(f_i))]))` evaluated as
 * relu(sum_i resize(W_i conv_i(f_i)) + b) -- the 1x1 fusion convolution and the bilinear resize commute.  chscale (NULL = none): fp32 [B, C]
 * factors >= 0 applied behind the activation -- the head's nn.Dropout2d ((U >= p) / (1 - p) per image and channel) without a pass of its own. */
int jg_resize_sum(int dtype, const void* x0, const void* x1, int H1, int W1, const void* x2, int H2, int W2, const void* x3, int H3, int W3,
                  void* y, int B, int Ho, int Wo, int C, int act, const float* chscale, jg_stream_t s);
/* Round 6: adjoint of jg_resize_sum in separable form: g = dy * act'(y) (written when `g` != NULL: the gradient of the full-resolution term),
 * dx_i = R_y^T (R_x^T g) for every lower-resolution term (NULL = not wanted); `ws`: jg_resize_sum_bwd_ws_floats(...) floats.  Two launches for the
 * activation-gradient pass + three gather launches of jg_bilinear_bwd (autograd of SegformerHead.forward's resize + sum);
 * chscale as in jg_resize_sum (y is then the scaled output).  JG_ERR_UNSUPPORTED when a row of g (Wo * C 16-bit values) exceeds 64 KB of LDS. */
int64_t jg_resize_sum_bwd_ws_floats(int B, int Ho, int C, int W1, int W2, int W3);
int jg_resize_sum_bwd(int dtype, const void* y, const void* dy, void* g, void* dx1, int H1, int W1, void* dx2, int H2, int W2, void* dx3, int H3,
                      int W3, float* ws, int B, int Ho, int Wo, int C, int act, const float* chscale, jg_stream_t s);
int jg_bn_coef(const float* sums, const float* gamma, const float* beta, float* running_mean, float* running_var, float* ab, float* mr, int B,
               int HW, int C, float eps, float momentum, int training, jg_stream_t s);
int jg_bn_bwd_coef(const float* red, const float* gamma, const float* mr, float* pqr, float* dgamma, float* dbeta, int B, int HW, int C,
                   int training, jg_stream_t s);
int jg_attn_compose_fwd(int dtype, const void* img, const void* logits, const void* xin, void* out, int B, int S, int f, int na, int ni, int nc,
                        int ldimg, int ldl, int ldx, int ldo, jg_stream_t s);
int jg_attn_compose_bwd(int dtype, const void* img, const void* logits, const void* xin, const void* dout, void* dimg, void* dlogits, void* dxin,
                        int B, int S, int f, int na, int ni, int nc, int ldimg, int ldl, int ldx, int ldo, jg_stream_t s);
int jg_scale(int dtype, const void* x, const float* scale, const void* res, void* y, int B, int64_t HW, int C, int per_channel, jg_stream_t s);

/* Projected discriminator (models/modules/projected_d/discriminator.py:13-77,166-286; blocks.py:11-13,182-200).
 * Spectral normalisation (torch.nn.utils.spectral_norm, one power iteration per training forward) of a convolution whose fp32 master
 * W is [Cout][R][S][Cin]:  v = normalize(W^T u), u = normalize(W v), sigma = u . W v  (u [Cout], v [Cin*RS] in the REFERENCE's
 * weight.view(Cout, -1) order, both updated in place; ws: >= RS*Cin + Cout + 2 floats);
 * jg_spectral_weights writes the 16-bit working copies of W / sigma (straight, and flipped + transposed for the input gradient);
 * jg_spectral_wgrad_fix turns the gradient w.r.t. W_sn into the gradient w.r.t. W (u, v constant) and ADDS it to g:
 *   g += (dWsn - <dWsn, W_sn> u v^T) / sigma          (ws: 1 float).
 * jg_hinge_loss: gan_mode "projected" (models/modules/loss.py:77-84): mode 0 mean relu(1 - p), 1 mean relu(1 + p), 2 mean(-p) over
 * the cvalid leading channels of an NHWC logit map; *loss += scale * value, dpred = grad_scale * scale * d value / d pred. */
int jg_spectral_power_iter(const float* W, float* u, float* v, float* sigma, float* ws, int Cout, int RS, int Cin, float eps, jg_stream_t s);
int jg_spectral_weights(int dtype, const float* W, const float* sigma, void* w16, void* w16T, int Cout, int RS, int Cin, int CoutP, int CinP,
                        jg_stream_t s);
int jg_spectral_wgrad_fix(const float* dWsn, const float* W, const float* u, const float* v, const float* sigma, float* g, float* ws, int Cout,
                          int RS, int Cin, jg_stream_t s);
/* The three steps above for ALL spectral-norm layers of a discriminator in table-driven launches (4 per forward instead of 5 per layer,
 * 2 per backward instead of 3 per layer).  `table`: device array of L jg_sn_desc records (pointers into the arena / the u, v buffers and
 * offsets into the caller's scratch buffers: fbuf fp32 = per layer ws [K + Cout + 2] (cleared by the call over the first zero_floats
 * floats) and the forward's snapshot sigma | u | v; hbuf 16-bit = w16 | w16T per layer; dbuf fp32 = dWsn per layer). */
typedef struct jg_sn_desc {
  const float* W; float* u; float* v; float* g;
  int64_t ws_off, snap_off, w16_off, dw_off;
  int32_t Cout, RS, Cin, CoutP, CinP, pad_;
} jg_sn_desc;
int jg_spectral_group_forward(int dtype, const void* table, int L, float* fbuf, int64_t zero_floats, void* hbuf, int max_K, int max_Cout,
                              int64_t max_w, float eps, jg_stream_t s);
int jg_spectral_group_wgrad_fix(const void* table, int L, const float* fbuf, const float* dbuf, float* dots, uint64_t mask, int64_t max_n,
                                jg_stream_t s);
int jg_hinge_loss(int dtype, const void* pred, float* loss, void* dpred, int64_t npix, int cpad, int cvalid, int mode, float scale,
                  float grad_scale, jg_stream_t s);

/* One DDPM ancestral sampling step after the UNet (DiffusionGenerator.p_sample / p_mean_variance, restoration_ddpm:
 * models/modules/diffusion_generator.py:187-284, predict_start_from_noise / q_posterior: diffusion_utils.py:122-137):
 *   y0_hat = clamp(sr*y_t - srm1*noise_hat, -1, 1);  y' = c1*y0_hat + c2*y_t + z*sigma;  y' = y_0*(1-m) + m*y'
 * coef[b][5] = {sqrt_recip_gammas[t], sqrt_recipm1_gammas[t], posterior_mean_coef1[t], posterior_mean_coef2[t],
 * exp(0.5*posterior_log_variance_clipped[t])}; z may be NULL (t == 0).  y_t fp32 NCHW is updated in place and the next
 * UNet input [y_cond | y' | 0] is written as 16-bit NHWC with Cpad_out channels.  clip_denoised bit 0 clamps y0_hat,
 * bit 1 clamps y' before the mask blend: the reference's DDIM step (ddim_p_mean_variance :383-455) is the same kernel
 * with coef = {0, -1, coef_eps - sqrt(g_prev (1 - g_t) / g_t), sqrt(g_prev / g_t), 0}, clip 3, z NULL. */
int jg_ddpm_p_sample(int dtype, float* y_t, const float* y_cond, const void* noise_hat, const float* z, const float* y_0,
                     const int64_t* mask, const float* coef, void* xin, int B, int C, int H, int W, int Cpad_in, int Cpad_out,
                     int clip_denoised, jg_stream_t s);

/* Consistency-model (cm_model) glue around the UNet, models/modules/cm_generator.py / models/cm_model.py.
 *   cm_noisy   : noisy = x + sigma[b]*noise, blended with clamp(mask,0,1) (forward :452-460); written as fp32 NCHW
 *                and as the 16-bit NHWC UNet input [cond | noisy | 0-pad] (= torch.cat([x_cond, x], 1) of :377-381)
 *   cm_combine : c_skip[b]*noisy + c_out[b]*F  (cm_forward :383-385), fp32 NCHW
 *   cm_loss    : lambda * mean(w[b] * pseudo_huber(mask*pred, mask*target)) with pred/target formed on the fly from
 *                the two UNet outputs, and dL/dF_next in the same pass (compute_cm_loss cm_model.py:353-375,
 *                pseudo_huber_loss :27-43; the mask multiplies AS IS, like the reference); loss accumulates (zero it)
 *   noise_level_embedding : [sin | cos](sigma * W * 2 pi)  (NoiseLevelEmbedding.forward :276-280) */
int jg_cm_noisy(int dtype, const float* x, const float* noise, const float* sigma, const int64_t* mask, const float* cond,
                float* out_nchw, void* out_nhwc, int B, int C, int Ccond, int H, int W, int Cpad, jg_stream_t s);
int jg_cm_combine(int dtype, const float* noisy, const void* F, const float* cskip, const float* cout, float* out, int B,
                  int C, int H, int W, int Cpad, jg_stream_t s);
int jg_cm_loss(int dtype, const void* Fn, const void* Fc, const float* noisy_n, const float* noisy_c, const float* cs_n,
               const float* co_n, const float* cs_c, const float* co_c, const int64_t* mask, const float* w, float* loss,
               void* dFn, int B, int C, int H, int W, int Cpad, float c_huber, float lambda, float grad_scale, jg_stream_t s);
int jg_noise_level_embedding(const float* sigma, const float* W, float* emb, int Bn, int half, jg_stream_t s);
/* gradient of the embedding with respect to W, ACCUMULATED into dW (the reference trains W: set_requires_grad(net, True),
 * base_model.py:1196-1217) */
int jg_noise_level_embedding_bwd(const float* sigma, const float* W, const float* demb, float* dW, int Bn, int half,
                                 jg_stream_t s);

/* Device-side input pipeline of the self-supervised inpainting datasets: per image crop window (oy, ox) of size S x S out of the
 * uint8 HWC source [B,H,W,3] (+ uint8 label mask [B,H,W], may be NULL), optional horizontal flip, ToTensor + Normalize(0.5, 0.5)
 * (data/base_dataset.py:513-528), ToTensorMask -> int64 [B,1,S,S] (:892-917) and fill_mask_with_random
 * (data/online_creation.py:1366-1376; data/self_supervised_labeled_mask_dataset.py:46-62):
 *   B = (img / 255 - 0.5) / 0.5;  A = B (1 - m) + noise m,  m = (mask != 0).   win = int32 [B][3] = (oy, ox, flip).
 * A, Bimg: fp32 NCHW [B,3,S,S]; noise: fp32 NCHW N(0,1) draws (NULL: treated as 0).  Bit-identical to the CPU transforms. */
int jg_input_pipeline(const uint8_t* img, const uint8_t* mask, const int32_t* win, const float* noise, float* A, float* Bimg,
                      int64_t* mask_out, int B, int H, int W, int S, jg_stream_t s);
/* `load_size` resize of the decoded uint8 batch in front of jg_input_pipeline (data/base_dataset.py:441-443: transforms.Resize(BICUBIC) on a
 * PIL image = PIL Image.resize; :749-763 ResizeMask: NEAREST for the label mask).  jg_resample_u8 = ONE separable pass of PIL's fixed-point
 * resampler (Pillow Resample.c: out = clip8((2^21 + sum_k in[lo + k] * kk[k]) >> 22)) over [B, H, W, 3] uint8: horizontal (vertical = 0,
 * Hin == Hout) or vertical (Win == Wout); bounds [out, 2] = (first source index, tap count), kk [out, ksize] = 22-bit fixed-point taps,
 * both computed by the caller as PIL's precompute_coeffs / normalize_coeffs_8bpc do.  jg_resize_nearest_u8: [B, Hin, Win] -> [B, Hout, Wout]
 * through the caller's source-index tables ytab [Hout] / xtab [Wout] (PIL accumulates the source coordinate incrementally in double). */
int jg_resample_u8(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk, int ksize, int vertical, int B, int Hin, int Win,
                   int Hout, int Wout, jg_stream_t s);
int jg_resize_nearest_u8(const uint8_t* in, uint8_t* out, const int32_t* ytab, const int32_t* xtab, int B, int Hin, int Win, int Hout, int Wout,
                         jg_stream_t s);

/* NHWC(T, Cpad) <-> NCHW(fp32, C) layout converters at the module boundary. */
int jg_nhwc_to_nchw_f32(int dtype, const void* x, float* y, int B, int C, int H, int W, int Cpad, jg_stream_t s);
int jg_nchw_f32_to_nhwc(int dtype, const float* x, void* y, int B, int C, int H, int W, int Cpad, jg_stream_t s);

/* Fused multi-tensor optimizer over a flat fp32 arena (train.py:51-62 torch.optim.AdamW/Adam;
 * BaseModel.compute_step + ema_step, models/base_model.py:1250-1297):
 *   g' = g * grad_scale (+ wd * p when !decoupled); p *= (1 - lr wd) when decoupled;
 *   m = b1 m + (1-b1) g'; v = b2 v + (1-b2) g'^2; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps);
 *   ema = p + ema_beta (ema - p)   (ema may be NULL);  g = 0 when zero_grad. */
int jg_adamw_ema(float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1, float beta2,
                 float eps, float wd, int decoupled, int step, float grad_scale, float ema_beta, int zero_grad,
                 jg_stream_t s);
/* The optimizer factory of train.py:51-62 as one entry point: kind 0 = torch.optim.Adam, 1 = AdamW (both = jg_adamw_ema),
 * 2 = torch.optim.RAdam (coupled weight decay, rectification term computed on the host from `step`),
 * 3 = Lion (util/lion_pytorch.py:60-82; uses m only, v may be NULL).  adam8bit (bitsandbytes) is not provided.
 * skip / nskipped (optional device ints): when *skip != 0 the step is dropped -- parameters, moments and EMA untouched, gradients
 * cleared, *nskipped += 1 -- the behaviour of torch.cuda.amp.GradScaler.step on non-finite gradients (models/base_model.py:1268-1274).
 * jg_grad_nonfinite sets *flag = 1 if any gradient element is inf / NaN (the caller clears the flag). */
int jg_optim_step(int kind, float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1, float beta2,
                  float eps, float wd, int step, float grad_scale, float ema_beta, int zero_grad, const int* skip, int* nskipped,
                  jg_stream_t s);
int jg_adamw_ema_skip(float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1, float beta2,
                      float eps, float wd, int decoupled, int step, float grad_scale, float ema_beta, int zero_grad, const int* skip,
                      int* nskipped, jg_stream_t s);
int jg_grad_nonfinite(const float* g, int64_t n, int* flag, jg_stream_t s);
/* Stand-alone EMA update ema = p + beta (ema - p) (BaseModel.ema_step, models/base_model.py:1284-1297) for the
 * micro-iterations of a gradient-accumulation cycle in which the optimizer does not step. */
int jg_ema_update(float* ema, const float* p, int64_t n, float beta, jg_stream_t s);
/* 16-bit working copies of the conv weights after an optimizer step.  One descriptor per layer
 * (device array of int64[8]): {src_off, dst_off, dstT_off, Cout, RS, Cin, CoutPad, CinPad}.
 *   w16 [CoutPad][R][S][CinPad]  = cast(p[Cout][R][S][Cin]) zero padded
 *   w16T[CinPad][R][S][CoutPad]  = w16[co][R-1-r][S-1-s][ci]   (input-gradient weights; dstT_off < 0 skips)
 * R, S taken from desc RS = R*65536 + S. */
int jg_refresh_weights(int dtype, const float* p, void* w16, void* w16T, const int64_t* desc, int nlayers, jg_stream_t s);

/* ---- ViT feature network of the projected discriminator (csrc/vit.hip) ------------------------------------------------------
 * D_proj_network_type "vitsmall" (examples/example_gan_mario2sonic.json): timm `vit_small_patch16_224` created at D_proj_interp, read
 * through `configure_get_feats_vit_timm` (models/modules/projected_d/projector.py:138-153,252-253,327-331).  Linear layers = jg_conv2d_nt,
 * LayerNorm forward = jg_layernorm_fwd.
 *   vit_attention_fwd/bwd : timm Attention.forward -- softmax((q * scale) k^T) v per head -- for token sequences of ANY length T (257 =
 *                       16 x 16 patches + class token), head_dim 32 or 64.  q / k / v: three views of the packed projection
 *                       ([B, T, ld] rows, head h at + h * head_stride elements; timm's qkv channel order is [3][heads][head_dim]:
 *                       q = qkv, k = qkv + C, v = qkv + 2C, ld = 3C, head_stride = head_dim).  o [B, T, heads * head_dim], L [B * heads, T]
 *                       logsumexp (saved for the backward).  bwd: dq / dk / dv views of the projection gradient (row stride ldd), Dq
 *                       [B * heads, T] scratch
 *   vit_tokens_fwd/bwd : cat(cls_token, patch_embed(x)) + pos_embed (projector.py:140-143); bwd = gradient of the patch embeddings
 *   gelu_fwd/bwd      : nn.GELU() (exact, erf) of timm's Mlp; bwd: dx = dy gelu'(x)
 *   layernorm_bwd_res : input gradient of a frozen nn.LayerNorm plus the residual branch of a pre-norm block: dx = res + LN'(dy) (res NULL = 0)
 *   (transpose2d      : below)
 *   unpatchify        : adjoint of the patch gather of PatchEmbed's Conv2d(k = s = P): dimg[b, ph P + r, pw P + s, ci] =
 *                       dpatch[(b, ph, pw)][ci P P + (P-1-r) P + (P-1-s)], 8 channels -- dpatch = dy x w16T (jg_refresh_weights layout) */
int jg_vit_attention_fwd(int dtype, const void* q, const void* k, const void* v, int64_t ld, int64_t head_stride, void* o, float* L, int B,
                         int T, int heads, int head_dim, float scale, jg_stream_t s);
int jg_vit_attention_bwd(int dtype, const void* q, const void* k, const void* v, int64_t ld, int64_t head_stride, const void* o, const float* L,
                         const void* d_o, void* dq, void* dk, void* dv, int64_t ldd, float* Dq, int B, int T, int heads, int head_dim,
                         float scale, jg_stream_t s);
int jg_vit_tokens_fwd(int dtype, const void* patch, const float* cls, const float* pos, void* y, int B, int N, int C, jg_stream_t s);
int jg_vit_tokens_bwd(int dtype, const void* dy, void* dpatch, int B, int N, int C, jg_stream_t s);
int jg_gelu_fwd(int dtype, const void* x, void* y, int64_t n, jg_stream_t s);
int jg_gelu_bwd(int dtype, const void* x, const void* dy, void* dx, int64_t n, jg_stream_t s);
int jg_layernorm_bwd_res(int dtype, const void* x, const void* dy, const float* gamma, const float* mr, const void* res, void* dx, int64_t R,
                         int C, jg_stream_t s);
int jg_unpatchify(int dtype, const void* dpatch, void* dimg, int B, int Hp, int Wp, int P, jg_stream_t s);
/* dst[b][c][r] = src[b][r][c], 16-bit elements: the `x.transpose(2, 1).contiguous()` of configure_get_feats_vit_timm (projector.py:148-150) in
 * front of nn.Flatten() of the MLP heads (discriminator.py:212), whose weights are laid out for [B, C * T]; its own adjoint */
int jg_transpose2d(int dtype, const void* src, void* dst, int B, int R, int C, jg_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* JG355_H */

"""`torch.ops.jg355.*` for the CUT family (VERDICT r4 missing #2; north_star: the kernels "exposed to the Python host as torch.ops"; precedent in
the reference: /root/reference/models/modules/op/upfirdn2d.py:19-167, a custom op with its own forward / backward pair).

Every op here is a torch.library custom op over the C ABI of include/jg355.h with a fake (meta) kernel and an autograd formula made of
other jg355 ops, so it traces (FakeTensor / torch.compile), composes with autograd and passes torch.library.opcheck
(tests/test_gpu_0_ops.py::test_torch_ops_surface_cut).  The ops are FUNCTIONAL: parameter gradients are RETURNED as fresh tensors (the
module graph's arena-accumulating nodes add them in place; `ops.torch_ops_boundary()` switches the modules to these ops, and
tests/test_gpu_5_cutloss.py::test_cut_step_through_torch_ops holds the two forms together on one CUT step).

  layer_norm / layer_norm_bwd                nn.LayerNorm of the SegFormer blocks and the ViT projector
  dwconv3x3_gelu / _bwd                      MixFFN's depth-wise 3x3 (+ GELU)
  attention_smallkv / _bwd                   EfficientMultiheadAttention core (spatially reduced keys / values)
  vit_attention / _bwd, gelu / gelu_bwd      timm Attention core / nn.GELU of the ViT projector
  reflect_pad2d / _bwd, reflect_conv2d / reflect_conv2d_wgrad      ReflectionPad2d(+ 3x3 conv in one launch) of the ResnetBlocks
  dilate2d                                   zero insertion (ConvTranspose2d = dilate + stride-1 conv on the flipped weights; stride-s dgrad)
  act / act_bwd                              stand-alone ReLU / LeakyReLU / Tanh
  gather_patches / scatter_patches, l2_normalize / _bwd, patch_nce / patch_nce_bwd      PatchSampleF + PatchNCE / MoNCE (Sinkhorn inside)
  gan_loss, hinge_loss                       GANLoss (lsgan / vanilla / wgangp) and the projected hinge; return (loss, d loss / d pred)
  spectral_weight / _bwd                     torch.nn.utils.spectral_norm: one power iteration, W / sigma; gradient through 1 / sigma
  bilinear2 / bilinear2_bwd                  F.interpolate(mode="bilinear", align_corners=...)
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check
from . import ops
from .ops import SINKHORN_ITERS, _DT, _dt, _p, _st, conv_nt, sgemm, wgrad_tn

T = torch.Tensor
op = torch.library.custom_op


def _z(like, dtype=torch.float32):
    return torch.zeros(0, device=like.device, dtype=dtype)


# ---- LayerNorm ---------------------------------------------------------------------------------------------------------------------
@op("jg355::layer_norm", mutates_args=())
def layer_norm(x: T, weight: T, bias: T, eps: float) -> Tuple[T, T]:
    """nn.LayerNorm(C, eps) over the last dimension of a 16-bit tensor -> (y, mr [rows, 2] = mean | rstd saved for the backward)"""
    x = x.contiguous()
    C = x.shape[-1]
    R = x.numel() // C
    y = torch.empty_like(x)
    mr = torch.empty((R, 2), device=x.device, dtype=torch.float32)
    check(_lib.lib().jg_layernorm_fwd(_dt(x), x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mr.data_ptr(), R, C, eps, _st()),
          "jg_layernorm_fwd")
    return y, mr


@layer_norm.register_fake
def _(x, weight, bias, eps):
    return torch.empty_like(x), x.new_empty((x.numel() // x.shape[-1], 2), dtype=torch.float32)


@op("jg355::layer_norm_bwd", mutates_args=())
def layer_norm_bwd(x: T, dy: T, weight: T, mr: T) -> Tuple[T, T, T]:
    x, dy = x.contiguous(), dy.contiguous()
    C = x.shape[-1]
    dx = torch.empty_like(x)
    dg, db = torch.zeros(C, device=x.device, dtype=torch.float32), torch.zeros(C, device=x.device, dtype=torch.float32)
    check(_lib.lib().jg_layernorm_bwd(_dt(x), x.data_ptr(), dy.data_ptr(), weight.data_ptr(), mr.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                      db.data_ptr(), x.numel() // C, C, _st()), "jg_layernorm_bwd")
    return dx, dg, db


@layer_norm_bwd.register_fake
def _(x, dy, weight, mr):
    C = x.shape[-1]
    return torch.empty_like(x), x.new_empty((C,), dtype=torch.float32), x.new_empty((C,), dtype=torch.float32)


def _ln_setup(ctx, inputs, output):
    x, weight, bias, eps = inputs
    ctx.save_for_backward(x, weight, output[1])


def _ln_backward(ctx, dy, dmr):
    x, weight, mr = ctx.saved_tensors
    dx, dg, db = torch.ops.jg355.layer_norm_bwd(x, dy, weight, mr)
    return dx, dg, db, None


layer_norm.register_autograd(_ln_backward, setup_context=_ln_setup)


# ---- depth-wise 3x3 (+ GELU) ---------------------------------------------------------------------------------------------------------
@op("jg355::dwconv3x3_gelu", mutates_args=())
def dwconv3x3_gelu(x: T, weight: T, bias: Optional[T], gelu: bool) -> Tuple[T, T]:
    """nn.Conv2d(C, C, 3, padding=1, groups=C) (+ nn.GELU()) on an NHWC map; weight fp32 [C, 1, 3, 3] -> (y, pre-activation; empty without gelu)"""
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    pre = torch.empty_like(x) if gelu else _z(x, x.dtype)
    w = weight.contiguous()
    check(_lib.lib().jg_dwconv3x3_fwd(_dt(x), x.data_ptr(), w.data_ptr(), _p(bias), pre.data_ptr() if gelu else None, y.data_ptr(), B, H, W, C,
                                      int(gelu), _st()), "jg_dwconv3x3_fwd")
    return y, pre


@dwconv3x3_gelu.register_fake
def _(x, weight, bias, gelu):
    return torch.empty_like(x), (torch.empty_like(x) if gelu else x.new_empty((0,)))


@op("jg355::dwconv3x3_gelu_bwd", mutates_args=())
def dwconv3x3_gelu_bwd(x: T, pre: T, dy: T, weight: T, gelu: bool) -> Tuple[T, T, T]:
    x, dy = x.contiguous(), dy.contiguous()
    B, H, W, C = x.shape
    du, dx = torch.empty_like(x), torch.empty_like(x)
    dw = torch.zeros((C, 1, 3, 3), device=x.device, dtype=torch.float32)
    db = torch.zeros(C, device=x.device, dtype=torch.float32)
    check(_lib.lib().jg_dwconv3x3_bwd(_dt(x), x.data_ptr(), pre.data_ptr() if gelu else None, dy.data_ptr(), weight.contiguous().data_ptr(),
                                      du.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), B, H, W, C, int(gelu), _st()), "jg_dwconv3x3_bwd")
    return dx, dw, db


@dwconv3x3_gelu_bwd.register_fake
def _(x, pre, dy, weight, gelu):
    C = x.shape[-1]
    return torch.empty_like(x), x.new_empty((C, 1, 3, 3), dtype=torch.float32), x.new_empty((C,), dtype=torch.float32)


def _dw_setup(ctx, inputs, output):
    x, weight, bias, gelu = inputs
    ctx.save_for_backward(x, output[1], weight)
    ctx.gelu, ctx.has_bias = gelu, bias is not None


def _dw_backward(ctx, dy, dpre):
    x, pre, weight = ctx.saved_tensors
    dx, dw, db = torch.ops.jg355.dwconv3x3_gelu_bwd(x, pre, dy, weight, ctx.gelu)
    return dx, dw, (db if ctx.has_bias else None), None


dwconv3x3_gelu.register_autograd(_dw_backward, setup_context=_dw_setup)


# ---- attention cores -------------------------------------------------------------------------------------------------------------------
@op("jg355::attention_smallkv", mutates_args=())
def attention_smallkv(q: T, kv: T, heads: int) -> Tuple[T, T]:
    """q [B, Tq, C], kv [B, Tkv, 2C] packed (k | v), head dim 32 -> (o [B, Tq, C], logsumexp [B, heads, Tq])"""
    q, kv = q.contiguous(), kv.contiguous()
    B, Tq, C = q.shape
    Tkv = kv.shape[1]
    o = torch.empty_like(q)
    lse = torch.empty((B, heads, Tq), device=q.device, dtype=torch.float32)
    es = q.element_size()
    check(_lib.lib().jg_attn_smallkv_fwd(_dt(q), q.data_ptr(), kv.data_ptr(), kv.data_ptr() + C * es, o.data_ptr(), lse.data_ptr(), B, Tq, Tkv, heads, C,
                                         2 * C, C, 1.0 / 32.0 ** 0.5, _st()), "jg_attn_smallkv_fwd")
    return o, lse


@attention_smallkv.register_fake
def _(q, kv, heads):
    return torch.empty_like(q), q.new_empty((q.shape[0], heads, q.shape[1]), dtype=torch.float32)


@op("jg355::attention_smallkv_bwd", mutates_args=())
def attention_smallkv_bwd(q: T, kv: T, o: T, lse: T, do: T, heads: int) -> Tuple[T, T]:
    q, kv, do = q.contiguous(), kv.contiguous(), do.contiguous()
    B, Tq, C = q.shape
    Tkv = kv.shape[1]
    dq = torch.empty_like(q)
    acc = torch.empty((2, B, Tkv, C), device=q.device, dtype=torch.float32)
    dkv = torch.empty_like(kv)
    es = q.element_size()
    check(_lib.lib().jg_attn_smallkv_bwd2(_dt(q), q.data_ptr(), kv.data_ptr(), kv.data_ptr() + C * es, o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                                          dq.data_ptr(), acc[0].data_ptr(), acc[1].data_ptr(), dkv.data_ptr(), dkv.data_ptr() + C * es, 2 * C, B, Tq, Tkv,
                                          heads, C, 2 * C, C, 1.0 / 32.0 ** 0.5, _st()), "jg_attn_smallkv_bwd2")
    return dq, dkv


@attention_smallkv_bwd.register_fake
def _(q, kv, o, lse, do, heads):
    return torch.empty_like(q), torch.empty_like(kv)


def _skv_setup(ctx, inputs, output):
    q, kv, heads = inputs
    ctx.save_for_backward(q, kv, output[0], output[1])
    ctx.heads = heads


def _skv_backward(ctx, do, dlse):
    q, kv, o, lse = ctx.saved_tensors
    dq, dkv = torch.ops.jg355.attention_smallkv_bwd(q, kv, o, lse, do, ctx.heads)
    return dq, dkv, None


attention_smallkv.register_autograd(_skv_backward, setup_context=_skv_setup)


@op("jg355::vit_attention", mutates_args=())
def vit_attention(qkv: T, heads: int) -> Tuple[T, T]:
    """timm Attention core on the packed projection [B, T, 3C] (channel order [3][heads][head_dim], head dim 32 / 64, any T) -> (a, logsumexp)"""
    from .modules.projected_d_vit import vit_attention_fwd

    return vit_attention_fwd(qkv.contiguous(), heads)


@vit_attention.register_fake
def _(qkv, heads):
    B, Tn, C3 = qkv.shape
    return qkv.new_empty((B, Tn, C3 // 3)), qkv.new_empty((B * heads, Tn), dtype=torch.float32)


@op("jg355::vit_attention_bwd", mutates_args=())
def vit_attention_bwd(qkv: T, a: T, lse: T, da: T, heads: int) -> T:
    from .modules.projected_d_vit import vit_attention_bwd as bwd

    return bwd(qkv.contiguous(), a, lse, da.contiguous(), heads)


@vit_attention_bwd.register_fake
def _(qkv, a, lse, da, heads):
    return torch.empty_like(qkv)


def _va_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[0], output[1])
    ctx.heads = inputs[1]


def _va_backward(ctx, da, dl):
    qkv, a, lse = ctx.saved_tensors
    return torch.ops.jg355.vit_attention_bwd(qkv, a, lse, da, ctx.heads), None


vit_attention.register_autograd(_va_backward, setup_context=_va_setup)


@op("jg355::gelu", mutates_args=())
def gelu(x: T) -> T:
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().jg_gelu_fwd(_dt(x), x.data_ptr(), y.data_ptr(), x.numel(), _st()), "jg_gelu_fwd")
    return y


@gelu.register_fake
def _(x):
    return torch.empty_like(x)


@op("jg355::gelu_bwd", mutates_args=())
def gelu_bwd(x: T, dy: T) -> T:
    x, dy = x.contiguous(), dy.contiguous()
    dx = torch.empty_like(x)
    check(_lib.lib().jg_gelu_bwd(_dt(x), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _st()), "jg_gelu_bwd")
    return dx


@gelu_bwd.register_fake
def _(x, dy):
    return torch.empty_like(x)


gelu.register_autograd(lambda ctx, dy: torch.ops.jg355.gelu_bwd(ctx.saved_tensors[0], dy),
                       setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


# ---- padding / dilation / activation ----------------------------------------------------------------------------------------------------
@op("jg355::reflect_pad2d", mutates_args=())
def reflect_pad2d(x: T, pad: int) -> T:
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty((B, H + 2 * pad, W + 2 * pad, C), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_reflect_pad2d(_dt(x), x.data_ptr(), y.data_ptr(), B, H, W, C, pad, _st()), "jg_reflect_pad2d")
    return y


@reflect_pad2d.register_fake
def _(x, pad):
    B, H, W, C = x.shape
    return x.new_empty((B, H + 2 * pad, W + 2 * pad, C))


@op("jg355::reflect_pad2d_bwd", mutates_args=())
def reflect_pad2d_bwd(dy: T, pad: int) -> T:
    """adjoint of ReflectionPad2d(pad): dy [B, H + 2 pad, W + 2 pad, C] folded back onto [B, H, W, C]"""
    dy = dy.contiguous()
    B, Hp, Wp, C = dy.shape
    H, W = Hp - 2 * pad, Wp - 2 * pad
    dx = torch.empty((B, H, W, C), device=dy.device, dtype=dy.dtype)
    check(_lib.lib().jg_reflect_pad2d_bwd(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, H, W, C, pad, _st()), "jg_reflect_pad2d_bwd")
    return dx


@reflect_pad2d_bwd.register_fake
def _(dy, pad):
    B, Hp, Wp, C = dy.shape
    return dy.new_empty((B, Hp - 2 * pad, Wp - 2 * pad, C))


reflect_pad2d.register_autograd(lambda ctx, dy: (torch.ops.jg355.reflect_pad2d_bwd(dy, ctx.pad), None),
                                setup_context=lambda ctx, inputs, output: setattr(ctx, "pad", inputs[1]))
reflect_pad2d_bwd.register_autograd(lambda ctx, g: (torch.ops.jg355.reflect_pad2d(g, ctx.pad), None),
                                    setup_context=lambda ctx, inputs, output: setattr(ctx, "pad", inputs[1]))


@op("jg355::dilate2d", mutates_args=())
def dilate2d(x: T, Ho: int, Wo: int, stride: int) -> T:
    """zero insertion: y[:, s i, s j] = x[:, i, j] in a [B, Ho, Wo, C] buffer"""
    return ops.dilate2d(x.contiguous(), Ho, Wo, stride)


@dilate2d.register_fake
def _(x, Ho, Wo, stride):
    return x.new_empty((x.shape[0], Ho, Wo, x.shape[-1]))


def _dil_setup(ctx, inputs, output):
    ctx.geo = (inputs[0].shape[1], inputs[0].shape[2], inputs[3])


def _dil_backward(ctx, dy):
    H, W, s = ctx.geo
    return dy[:, ::s, ::s][:, :H, :W].contiguous(), None, None, None       # the adjoint of zero insertion is a strided view


dilate2d.register_autograd(_dil_backward, setup_context=_dil_setup)


@op("jg355::act", mutates_args=())
def act(x: T, kind: int) -> T:
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().jg_act_fwd(_dt(x), x.data_ptr(), y.data_ptr(), x.numel(), kind, _st()), "jg_act_fwd")
    return y


@act.register_fake
def _(x, kind):
    return torch.empty_like(x)


@op("jg355::act_bwd", mutates_args=())
def act_bwd(y: T, dy: T, kind: int) -> T:
    y, dy = y.contiguous(), dy.contiguous()
    dx = torch.empty_like(y)
    check(_lib.lib().jg_act_bwd(_dt(y), y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), kind, _st()), "jg_act_bwd")
    return dx


@act_bwd.register_fake
def _(y, dy, kind):
    return torch.empty_like(y)


def _act_setup(ctx, inputs, output):
    ctx.save_for_backward(output)
    ctx.kind = inputs[1]


act.register_autograd(lambda ctx, dy: (torch.ops.jg355.act_bwd(ctx.saved_tensors[0], dy, ctx.kind), None), setup_context=_act_setup)


# ---- ReflectionPad2d(1) + 3x3 convolution in one launch ------------------------------------------------------------------------------------
@op("jg355::reflect_conv2d", mutates_args=())
def reflect_conv2d(x: T, w: T, bias: Optional[T]) -> T:
    """conv3x3(reflection_pad(x, 1)): x [B, H, W, Cin] 16-bit (Cin, Cout % 64 == 0, H, W % 16 == 0), w [Cout, 3, 3, Cin] (fp32 master or 16-bit)"""
    x = x.contiguous()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    y = torch.empty((B, H, W, Cout), device=x.device, dtype=x.dtype)
    conv_nt(x, w.to(x.dtype).contiguous(), y, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=3, S=3, pad=1, stride=1, Ho=H, Wo=W, ldx=Cin, ldw=9 * Cin,
            ldy=Cout, bias=bias, pad_mode=1)
    return y


@reflect_conv2d.register_fake
def _(x, w, bias):
    return x.new_empty((*x.shape[:3], w.shape[0]))


@op("jg355::reflect_conv2d_wgrad", mutates_args=())
def reflect_conv2d_wgrad(dy: T, x: T) -> Tuple[T, T]:
    dy, x = dy.contiguous(), x.contiguous()
    B, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    dw = torch.zeros((Cout, 3, 3, Cin), device=x.device, dtype=torch.float32)
    db = torch.zeros((Cout,), device=x.device, dtype=torch.float32)
    wgrad_tn(dy, x, dw, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=3, S=3, pad=1, stride=1, Ho=H, Wo=W, lddy=Cout, ldx=Cin, lddw=9 * Cin, dbias=db,
             splitk=ops._wgrad_splitk(((Cout + 127) // 128) * ((9 * Cin + 127) // 128), B * H * W), pad_mode=1)
    return dw, db


@reflect_conv2d_wgrad.register_fake
def _(dy, x):
    return x.new_empty((dy.shape[-1], 3, 3, x.shape[-1]), dtype=torch.float32), x.new_empty((dy.shape[-1],), dtype=torch.float32)


def _rc_setup(ctx, inputs, output):
    x, w, bias = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None


def _rc_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = dw = db = None
    if ctx.needs_input_grad[0]:      # full convolution over the padded domain, folded back by the reflection's adjoint
        wT = w.to(dy.dtype).permute(3, 1, 2, 0).flip(1, 2).contiguous()
        dx = torch.ops.jg355.reflect_pad2d_bwd(torch.ops.jg355.conv2d_nt(dy, wT, None, None, 2, 1, 1.0, 0.0), 1)
    if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
        dwf, dbf = torch.ops.jg355.reflect_conv2d_wgrad(dy, x)
        dw = dwf.to(w.dtype) if ctx.needs_input_grad[1] else None
        db = dbf if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    return dx, dw, db


reflect_conv2d.register_autograd(_rc_backward, setup_context=_rc_setup)


# ---- PatchSampleF / PatchNCE ------------------------------------------------------------------------------------------------------------
@op("jg355::gather_patches", mutates_args=())
def gather_patches(feat: T, ids: T, C: int) -> T:
    """feat [B, H, W, ld] 16-bit -> [B * P, C] fp32 rows at the flattened positions ids (int64 [P], shared by the batch)"""
    feat, ids = feat.contiguous(), ids.contiguous()
    B, H, W, ld = feat.shape
    P = ids.numel()
    out = torch.empty((B * P, C), device=feat.device, dtype=torch.float32)
    check(_lib.lib().jg_gather_rows(_dt(feat), feat.data_ptr(), ld, ids.data_ptr(), out.data_ptr(), B, H * W, C, P, _st()), "jg_gather_rows")
    return out


@gather_patches.register_fake
def _(feat, ids, C):
    return feat.new_empty((feat.shape[0] * ids.numel(), C), dtype=torch.float32)


@op("jg355::scatter_patches", mutates_args=())
def scatter_patches(dout: T, ids: T, B: int, H: int, W: int, ld: int, fp16: bool) -> T:
    dout, ids = dout.contiguous(), ids.contiguous()
    dt = torch.float16 if fp16 else torch.bfloat16
    dfeat = torch.zeros((B, H, W, ld), device=dout.device, dtype=dt)
    check(_lib.lib().jg_scatter_rows(_DT[dt], dfeat.data_ptr(), ld, ids.data_ptr(), dout.data_ptr(), B, H * W, dout.shape[1], ids.numel(), _st()),
          "jg_scatter_rows")
    return dfeat


@scatter_patches.register_fake
def _(dout, ids, B, H, W, ld, fp16):
    return dout.new_empty((B, H, W, ld), dtype=torch.float16 if fp16 else torch.bfloat16)


def _gp_setup(ctx, inputs, output):
    feat, ids, C = inputs
    ctx.save_for_backward(ids)
    ctx.geo = (*feat.shape, feat.dtype == torch.float16)


def _gp_backward(ctx, dout):
    (ids,) = ctx.saved_tensors
    B, H, W, ld, fp16 = ctx.geo
    return torch.ops.jg355.scatter_patches(dout, ids, B, H, W, ld, fp16), None, None


gather_patches.register_autograd(_gp_backward, setup_context=_gp_setup)


@op("jg355::l2_normalize", mutates_args=())
def l2_normalize(x: T, eps: float) -> Tuple[T, T]:
    x = x.contiguous()
    R, D = x.shape
    y = torch.empty_like(x)
    nrm = torch.empty(R, device=x.device, dtype=torch.float32)
    check(_lib.lib().jg_l2norm_fwd(x.data_ptr(), y.data_ptr(), nrm.data_ptr(), R, D, eps, _st()), "jg_l2norm_fwd")
    return y, nrm


@l2_normalize.register_fake
def _(x, eps):
    return torch.empty_like(x), x.new_empty((x.shape[0],))


@op("jg355::l2_normalize_bwd", mutates_args=())
def l2_normalize_bwd(y: T, nrm: T, dy: T, eps: float) -> T:
    dy = dy.contiguous()
    dx = torch.empty_like(y)
    check(_lib.lib().jg_l2norm_bwd(y.data_ptr(), nrm.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.shape[0], y.shape[1], eps, _st()), "jg_l2norm_bwd")
    return dx


@l2_normalize_bwd.register_fake
def _(y, nrm, dy, eps):
    return torch.empty_like(y)


def _l2_setup(ctx, inputs, output):
    ctx.save_for_backward(output[0], output[1])
    ctx.eps = inputs[1]


l2_normalize.register_autograd(lambda ctx, dy, dn: (torch.ops.jg355.l2_normalize_bwd(ctx.saved_tensors[0], ctx.saved_tensors[1], dy, ctx.eps), None),
                               setup_context=_l2_setup)


@op("jg355::patch_nce", mutates_args=())
def patch_nce(q: T, k: T, nimg: int, temp: float, pm1: float, monce: bool) -> Tuple[T, T, T, T, T]:
    """per-patch PatchNCE / MoNCE loss (base_NCE.py:17-66, monce.py:16-33; Sinkhorn iterations inside): q, k [nimg * P, D] fp32 L2-normalised
    -> (loss [nimg * P], logits S, and the optimal-transport state K, u-history, v-history: empty without monce)"""
    q, k = q.contiguous(), k.contiguous()
    R, D = q.shape
    P = R // nimg
    L = _lib.lib()
    S = torch.empty((nimg, P, P), device=q.device, dtype=torch.float32)
    sgemm(q, k, S, P, P, D, (D, 1), (D, 1), (P, 1), nimg, (P * D, P * D, P * P))
    loss = torch.empty(R, device=q.device, dtype=torch.float32)
    K, uh, vh = _z(q), _z(q), _z(q)
    u = v = None
    if monce:
        K = torch.empty_like(S)
        uh = torch.empty((nimg, SINKHORN_ITERS, P), device=q.device, dtype=torch.float32)
        vh = torch.empty((nimg, SINKHORN_ITERS + 1, P), device=q.device, dtype=torch.float32)
        check(L.jg_nce_sinkhorn_fwd(S.data_ptr(), K.data_ptr(), uh.data_ptr(), vh.data_ptr(), nimg, P, SINKHORN_ITERS, 1.0, _st()), "jg_nce_sinkhorn_fwd")
        u, v = uh[:, -1], vh[:, -1]
    check(L.jg_nce_ce(S.data_ptr(), _p(u), SINKHORN_ITERS * P, _p(v), (SINKHORN_ITERS + 1) * P, loss.data_ptr(), None, None, nimg, P, temp, pm1, None,
                      None, 1.0, _st()), "jg_nce_ce")
    return loss, S, K, uh, vh


@patch_nce.register_fake
def _(q, k, nimg, temp, pm1, monce):
    R = q.shape[0]
    P = R // nimg
    e = q.new_empty((0,))
    if monce:
        return q.new_empty((R,)), q.new_empty((nimg, P, P)), q.new_empty((nimg, P, P)), q.new_empty((nimg, SINKHORN_ITERS, P)), q.new_empty((nimg, SINKHORN_ITERS + 1, P))
    return q.new_empty((R,)), q.new_empty((nimg, P, P)), e, q.new_empty((0,)), q.new_empty((0,))


@op("jg355::patch_nce_bwd", mutates_args=())
def patch_nce_bwd(q: T, k: T, S: T, K: T, uh: T, vh: T, dloss: T, nimg: int, temp: float, pm1: float, monce: bool) -> Tuple[T, T]:
    L = _lib.lib()
    R, D = q.shape
    P = R // nimg
    dloss = dloss.contiguous().float()
    dS = torch.empty_like(S)
    gW = torch.empty_like(S) if monce else None
    gpos = torch.empty(R, device=q.device, dtype=torch.float32)
    scratch = torch.empty_like(q)
    u = uh[:, -1] if monce else None
    v = vh[:, -1] if monce else None
    check(L.jg_nce_ce(S.data_ptr(), _p(u), SINKHORN_ITERS * P, _p(v), (SINKHORN_ITERS + 1) * P, scratch.data_ptr(), dS.data_ptr(), _p(gW), nimg, P, temp, pm1,
                      dloss.data_ptr(), gpos.data_ptr(), 1.0, _st()), "jg_nce_ce")
    dk = torch.empty_like(k)
    sgemm(dS, q, dk, P, D, P, (1, P), (1, D), (D, 1), nimg, (P * P, P * D, P * D))
    if monce:
        dsh, drh = torch.empty_like(uh), torch.empty_like(uh)
        check(L.jg_nce_sinkhorn_bwd(K.data_ptr(), uh.data_ptr(), vh.data_ptr(), gW.data_ptr(), dsh.data_ptr(), drh.data_ptr(), dS.data_ptr(), nimg, P,
                                    SINKHORN_ITERS, _st()), "jg_nce_sinkhorn_bwd")
    dq = torch.empty_like(q)
    sgemm(dS, k, dq, P, D, P, (P, 1), (1, D), (D, 1), nimg, (P * P, P * D, P * D))
    check(L.jg_row_axpy(dq.data_ptr(), gpos.data_ptr(), k.data_ptr(), R, D, _st()), "jg_row_axpy")
    return dq, dk


@patch_nce_bwd.register_fake
def _(q, k, S, K, uh, vh, dloss, nimg, temp, pm1, monce):
    return torch.empty_like(q), torch.empty_like(k)


def _nce_setup(ctx, inputs, output):
    q, k, nimg, temp, pm1, monce = inputs
    ctx.save_for_backward(q, k, *output[1:])
    ctx.cfg = (nimg, temp, pm1, monce)


def _nce_backward(ctx, dloss, *unused):
    q, k, S, K, uh, vh = ctx.saved_tensors
    dq, dk = torch.ops.jg355.patch_nce_bwd(q, k, S, K, uh, vh, dloss, *ctx.cfg)
    return dq, dk, None, None, None, None


patch_nce.register_autograd(_nce_backward, setup_context=_nce_setup)


# ---- GAN objectives: (loss, d loss / d pred) -------------------------------------------------------------------------------------------------
@op("jg355::gan_loss", mutates_args=())
def gan_loss(pred: T, mode: int, target: float, scale: float) -> Tuple[T, T]:
    """GANLoss: mode 0 lsgan, 1 vanilla (BCE with logits), 2 wgangp, on an NHWC logit map whose channel 0 is valid"""
    pred = pred.contiguous()
    cpad = pred.shape[-1]
    loss = torch.zeros((), device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    check(_lib.lib().jg_gan_loss(_dt(pred), mode, pred.data_ptr(), target, loss.data_ptr(), dpred.data_ptr(), pred.numel() // cpad, cpad, scale, 1.0, _st()),
          "jg_gan_loss")
    return loss, dpred


@gan_loss.register_fake
def _(pred, mode, target, scale):
    return pred.new_empty((), dtype=torch.float32), torch.empty_like(pred)


@op("jg355::hinge_loss", mutates_args=())
def hinge_loss(pred: T, mode: int, scale: float) -> Tuple[T, T]:
    """GANLoss("projected"): mode 0 mean relu(1 - p), 1 mean relu(1 + p), 2 mean(-p) over every element"""
    pred = pred.contiguous()
    loss = torch.zeros((), device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    check(_lib.lib().jg_hinge_loss(_dt(pred), pred.data_ptr(), loss.data_ptr(), dpred.data_ptr(), pred.numel(), 1, 1, mode, scale, 1.0, _st()), "jg_hinge_loss")
    return loss, dpred


@hinge_loss.register_fake
def _(pred, mode, scale):
    return pred.new_empty((), dtype=torch.float32), torch.empty_like(pred)


def _loss_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _loss_backward_n(n):
    def bwd(ctx, g, gd):
        (dpred,) = ctx.saved_tensors
        return ((dpred.float() * g).to(dpred.dtype),) + (None,) * n
    return bwd


gan_loss.register_autograd(_loss_backward_n(3), setup_context=_loss_setup)
hinge_loss.register_autograd(_loss_backward_n(2), setup_context=_loss_setup)


# ---- spectral normalisation of a convolution weight -------------------------------------------------------------------------------------------
@op("jg355::spectral_weight", mutates_args=())
def spectral_weight(weight: T, u: T, v: T, training: bool) -> Tuple[T, T, T, T]:
    """torch.nn.utils.spectral_norm on W [Cout, R, S, Cin] fp32 (physical layout of the arena): one power iteration when `training`
    (returned as NEW u, v: the module copies them into its buffers), sigma = u . W v, W / sigma -> (W_sn, u', v', sigma)"""
    W = weight.contiguous()
    Cout = W.shape[0]
    K = W.numel() // Cout
    RS = W.shape[1] * W.shape[2] if W.dim() == 4 else 1
    un, vn = u.clone(), v.clone()
    sigma = torch.empty(1, device=W.device, dtype=torch.float32)
    if training:
        ws = torch.empty(K + Cout + 2, device=W.device, dtype=torch.float32)
        check(_lib.lib().jg_spectral_power_iter(W.data_ptr(), un.data_ptr(), vn.data_ptr(), sigma.data_ptr(), ws.data_ptr(), Cout, RS, K // RS, 1e-12, _st()),
              "jg_spectral_power_iter")
    else:
        sigma.copy_(torch.dot(un, W.reshape(Cout, K) @ vn).reshape(1))
    return W / sigma, un, vn, sigma


@spectral_weight.register_fake
def _(weight, u, v, training):
    return torch.empty_like(weight), torch.empty_like(u), torch.empty_like(v), weight.new_empty((1,))


@op("jg355::spectral_weight_bwd", mutates_args=())
def spectral_weight_bwd(dwsn: T, weight: T, u: T, v: T, sigma: T) -> T:
    """gradient w.r.t. W of W / sigma(W) with u, v constant: (dWsn - (u^T dWsn v) u v^T ... ) / sigma (jg_spectral_wgrad_fix)"""
    W = weight.contiguous()
    Cout = W.shape[0]
    K = W.numel() // Cout
    RS = W.shape[1] * W.shape[2] if W.dim() == 4 else 1
    g = torch.zeros_like(W)
    ws = torch.empty(1, device=W.device, dtype=torch.float32)
    check(_lib.lib().jg_spectral_wgrad_fix(dwsn.contiguous().float().data_ptr(), W.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(), g.data_ptr(),
                                           ws.data_ptr(), Cout, RS, K // RS, _st()), "jg_spectral_wgrad_fix")
    return g


@spectral_weight_bwd.register_fake
def _(dwsn, weight, u, v, sigma):
    return torch.empty_like(weight)


def _sw_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[1], output[2], output[3])


def _sw_backward(ctx, dwsn, du, dv, ds):
    W, un, vn, sigma = ctx.saved_tensors
    return torch.ops.jg355.spectral_weight_bwd(dwsn, W, un, vn, sigma), None, None, None


spectral_weight.register_autograd(_sw_backward, setup_context=_sw_setup)


# ---- bilinear resize ---------------------------------------------------------------------------------------------------------------------
@op("jg355::bilinear2", mutates_args=())
def bilinear2(x: T, Ho: int, Wo: int, align_corners: bool) -> T:
    x = x.contiguous()
    B, H, W, C = x.shape
    y = torch.empty((B, Ho, Wo, C), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_bilinear2_fwd(_dt(x), x.data_ptr(), y.data_ptr(), B, H, W, C, Ho, Wo, C, int(align_corners), _st()), "jg_bilinear2_fwd")
    return y


@bilinear2.register_fake
def _(x, Ho, Wo, align_corners):
    return x.new_empty((x.shape[0], Ho, Wo, x.shape[-1]))


@op("jg355::bilinear2_bwd", mutates_args=())
def bilinear2_bwd(dy: T, H: int, W: int, align_corners: bool) -> T:
    dy = dy.contiguous()
    B, Ho, Wo, C = dy.shape
    dx = torch.empty((B, H, W, C), device=dy.device, dtype=dy.dtype)
    check(_lib.lib().jg_bilinear2_bwd(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, H, W, C, Ho, Wo, C, int(align_corners), _st()), "jg_bilinear2_bwd")
    return dx


@bilinear2_bwd.register_fake
def _(dy, H, W, align_corners):
    return dy.new_empty((dy.shape[0], H, W, dy.shape[-1]))


def _bil_setup(ctx, inputs, output):
    ctx.geo = (inputs[0].shape[1], inputs[0].shape[2], inputs[3])


bilinear2.register_autograd(lambda ctx, dy: (torch.ops.jg355.bilinear2_bwd(dy, *ctx.geo), None, None, None), setup_context=_bil_setup)


# ---- compositions used by the boundary mode (no new kernels: autograd chains the ops above) -------------------------------------------------
def conv_transpose2d_via_ops(x, m, output_padding=0):
    """nn.ConvTranspose2d on jg355 ops: zero insertion + stride-1 convolution with the flipped / transposed MASTER weight (differentiable views:
    autograd carries the weight gradient back through flip / permute)"""
    B, H, W, Cin_t = x.shape
    Ho = (H - 1) * m.stride - 2 * m.pad + m.R + output_padding
    Wo = (W - 1) * m.stride - 2 * m.pad + m.S + output_padding
    Hd, Wd = Ho + 2 * m.pad - m.R + 1, Wo + 2 * m.pad - m.S + 1
    xd = torch.ops.jg355.dilate2d(x, Hd, Wd, m.stride)
    # module weight [Cin_t, Cout_t, k, k] -> forward conv weight [Cout_t][R][S][Cin_t] = w[ci_t][co_t][R-1-r][S-1-s]
    wf = m.weight.permute(1, 2, 3, 0).flip(1, 2)
    return torch.ops.jg355.conv2d_nt(xd, wf, m.bias, None, m.R - 1 - m.pad, 1, 1.0, 0.0)

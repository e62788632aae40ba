"""Projected discriminator on the HIP ops: mirror of /root/reference/models/modules/projected_d/discriminator.py (`SingleDisc` :13-77,
`MultiScaleD` :166-230, `ProjectedDiscriminator` :233-286), projector.py (`Proj` :490-589, `_make_scratch_ccm/_csm` :15-48,
`_make_efficientnet` :51-59) and blocks.py (`conv2d` = spectral_norm(nn.Conv2d) :11-13, `NormLayer` :28-32, `DownBlock` :182-200,
`FeatureFusionBlockMatrix` :248-287) for the convolutional ("efficientnet") variant, `proj_type` 2, `cout` 64, `expand` True.

What is and is not here
  * The frozen feature network's BACKBONE in the reference is a pretrained timm model (`tf_efficientnet_lite0`).  Round 3: its
    ARCHITECTURE is built (16 MBConv blocks, `make_efficientnet_lite0`, state_dict keys of timm re-homed by `_make_efficientnet`), on the
    depth-wise / point-wise kernels of csrc/effnet.hip; timm and its weights are not available offline, so the weights are random until a
    checkpoint is loaded (`jg_projd_pretrained`, or a reference D checkpoint) and a loud warning says so.  `backbone="standin"`
    (opt-in, tests) keeps the round-2 STAND-IN -- four stages at strides 4 / 8 / 16 / 32 with the same feature widths (24 / 40 / 112 /
    320) -- which, handed to the UNMODIFIED reference through a stubbed `timm.create_model`, produced tests/golden/projd.pt.  Everything downstream of the backbone is the reference's
    arithmetic: cross-channel mixing (1x1 convs), cross-scale mixing (add -> bilinear x2, align_corners -> 1x1 conv, top-down), the
    four spectral-norm / GroupNorm / LeakyReLU mini-discriminators, the concatenated logits, and the hinge objective.
  * state_dict() keys follow the reference (`freeze_feature_network.scratch.layer0_ccm.weight`,
    `discriminator.mini_discs.0.main.0.main.0.weight_orig / weight_u / weight_v`, ...).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .._autograd import JGFunction
from .. import _lib, ops
from .._lib import check
from ..ops import JG_ACT_LRELU, _dt, _p, _st, conv_nt, wgrad_tn
from .layers import JGConv2d, JGConvNd

TF_EFFICIENTNET_LITE0_WIDTHS = (24, 40, 112, 320)


# ---------------------------------------------------------------------------------------------------------------------
# small autograd nodes
# ---------------------------------------------------------------------------------------------------------------------
class _Bilinear2Fn(JGFunction):
    """F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=align) on an NHWC map (up-sampling)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo, align):
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty((B, Ho, Wo, C), device=x.device, dtype=x.dtype)
        check(_lib.lib().jg_bilinear2_fwd(_dt(x), x.data_ptr(), y.data_ptr(), B, H, W, C, Ho, Wo, C, int(align), _st()), "jg_bilinear2_fwd")
        ctx.geo = (H, W, Ho, Wo, int(align))
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        H, W, Ho, Wo, align = ctx.geo
        dy = dy.contiguous()
        B, C = dy.shape[0], dy.shape[-1]
        dx = torch.empty((B, H, W, C), device=dy.device, dtype=dy.dtype)
        check(_lib.lib().jg_bilinear2_bwd(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, H, W, C, Ho, Wo, C, align, _st()), "jg_bilinear2_bwd")
        return dx, None, None, None


def bilinear(x, Ho, Wo, align_corners):
    return _Bilinear2Fn.apply(x, Ho, Wo, align_corners)


class _AddFn(JGFunction):
    @staticmethod
    def forward(ctx, a, b):
        return ops.axpby(a.contiguous(), 1.0, b.contiguous(), 1.0)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class _HingeFn(JGFunction):
    @staticmethod
    def forward(ctx, pred, mode, scale):
        pred = pred.contiguous()
        loss = torch.zeros((), device=pred.device, dtype=torch.float32)
        dpred = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        check(_lib.lib().jg_hinge_loss(_dt(pred), pred.data_ptr(), loss.data_ptr(), _p(dpred), pred.numel(), 1, 1, mode, scale, 1.0, _st()),
              "jg_hinge_loss")
        ctx.dpred = dpred
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        dpred, ctx.dpred = ctx.dpred, None
        return (dpred.float() * g).to(dpred.dtype), None, None          # a few hundred logits: not worth a launch of its own


def hinge_loss(pred, target_is_real, relu=True, scale=1.0):
    """GANLoss("projected") (models/modules/loss.py:77-84) on a logits tensor whose every element is valid:
    relu: mean relu(1 - p) (real) / mean relu(1 + p) (fake);  not relu (generator): mean(-p)."""
    mode = (0 if target_is_real else 1) if relu else 2
    return _HingeFn.apply(pred, mode, float(scale))


# ---------------------------------------------------------------------------------------------------------------------
# all spectral-norm layers of a discriminator in table-driven launches (csrc/projected_d.hip, jg_spectral_group_*)
# ---------------------------------------------------------------------------------------------------------------------
SN_GROUPED = os.environ.get("JG_SN_GROUPED", "1") != "0"


class _SnGroup:
    """the static side: the 14 SpectralConv2d modules of a ProjectedDiscriminator, their descriptor table on the device (pointers into
    the arena and the u / v buffers, offsets into per-forward scratch) and the scratch sizes"""

    def __init__(self, mods, dtype):
        import ctypes as C
        import numpy as np

        self.mods, self.dtype, self.L = mods, dtype, len(mods)
        assert self.L <= 64
        dev = mods[0].weight_orig.device
        ws = snap = w16 = dw = 0
        rec = []
        for mod in mods:
            m = mod.meta
            K, RS = m.R * m.S * m.Cin_real, m.R * m.S
            rec.append(dict(ws=ws, snap=0, w16=w16, dw=dw, Cout=m.Cout_real, RS=RS, Cin=m.Cin_real, CoutP=m.Cout, CinP=m.Cin, K=K))
            ws += K + m.Cout_real + 2
            w16 += 2 * m.Cout * RS * m.Cin
            dw += m.Cout_real * K
        self.zero_floats = ws
        for r in rec:                       # snapshots behind the (cleared) workspaces
            r["snap"] = ws + snap
            snap += 1 + r["Cout"] + r["K"]
        self.fbuf_floats, self.hbuf_elems, self.dbuf_floats = ws + snap, w16, dw + self.L      # dbuf: dWsn of every layer | L dot products
        self.dots_off = dw
        self.rec = rec
        self.max_K = max(r["K"] for r in rec)
        self.max_Cout = max(r["Cout"] for r in rec)
        self.max_w = max(r["CoutP"] * r["RS"] * r["CinP"] for r in rec)
        self.max_n = max(r["Cout"] * r["K"] for r in rec)
        tab = np.zeros((self.L, 11), dtype=np.int64)         # 8 x 8-byte fields + 6 x int32 = 88 bytes per record
        ints = tab.view(np.int32).reshape(self.L, 22)
        self.ptrs = []
        for i, (mod, r) in enumerate(zip(mods, rec)):
            g = mod.weight_orig.grad
            if g is None:
                raise RuntimeError("spectral conv weight has no arena-backed .grad")
            p = (mod.weight_orig.data_ptr(), mod.weight_u.data_ptr(), mod.weight_v.data_ptr(), g.data_ptr())
            self.ptrs.append(p)
            tab[i, 0:4] = p
            tab[i, 4:8] = (r["ws"], r["snap"], r["w16"], r["dw"])
            ints[i, 16:22] = (r["Cout"], r["RS"], r["Cin"], r["CoutP"], r["CinP"], 0)
        self.table = torch.from_numpy(tab).to(dev)

    def still_valid(self):
        return all(p == (m.weight_orig.data_ptr(), m.weight_u.data_ptr(), m.weight_v.data_ptr(), m.weight_orig.grad.data_ptr())
                   for p, m in zip(self.ptrs, self.mods))


class _SnPass:
    """one forward of the discriminator: power iteration + working copies of every layer (4 launches + 1 fill), and -- when the weights take
    a gradient -- the dWsn buffers of its backward, fixed up by 2 launches once every layer has delivered"""

    def __init__(self, group):
        L = _lib.lib()
        self.g = g = group
        dev = g.table.device
        self.fbuf = torch.empty(g.fbuf_floats, device=dev, dtype=torch.float32)
        self.hbuf = torch.empty(g.hbuf_elems, device=dev, dtype=g.dtype)
        check(L.jg_spectral_group_forward(_lib.JG_F16 if g.dtype == torch.float16 else _lib.JG_BF16, g.table.data_ptr(), g.L, self.fbuf.data_ptr(),
                                          g.zero_floats, self.hbuf.data_ptr(), g.max_K, g.max_Cout, g.max_w, 1e-12, _st()), "jg_spectral_group_forward")
        self.dbuf, self.mask, self.flushed = None, 0, False

    def views(self, i):
        r = self.g.rec[i]
        n = r["CoutP"] * r["RS"] * r["CinP"]
        w16 = self.hbuf[r["w16"]:r["w16"] + n].view(r["CoutP"], -1, r["CinP"])
        w16T = self.hbuf[r["w16"] + n:r["w16"] + 2 * n].view(r["CinP"], -1, r["CoutP"])
        return w16, w16T

    def dwsn(self, i):
        if self.dbuf is None:
            self.dbuf = torch.zeros(self.g.dbuf_floats, device=self.fbuf.device, dtype=torch.float32)
        r = self.g.rec[i]
        return self.dbuf[r["dw"]:r["dw"] + r["Cout"] * r["K"]].view(r["Cout"], r["K"])

    def delivered(self, i):
        self.mask |= 1 << i
        if self.mask == (1 << self.g.L) - 1:
            self.flush()

    def flush(self):
        if self.flushed or not self.mask:
            return
        g = self.g
        if ops.WGRAD_DEFER:         # inside ops.deferred_wgrads(): the dWsn this fix reads are still queued -- issue them (grouped) first
            ops.flush_deferred_wgrads()
        check(_lib.lib().jg_spectral_group_wgrad_fix(g.table.data_ptr(), g.L, self.fbuf.data_ptr(), self.dbuf.data_ptr(),
                                                     self.dbuf.data_ptr() + 4 * g.dots_off, self.mask, g.max_n, _st()), "jg_spectral_group_wgrad_fix")
        self.flushed = True


# ---------------------------------------------------------------------------------------------------------------------
# spectral-norm convolution (blocks.py:11-13)
# ---------------------------------------------------------------------------------------------------------------------
class _SpectralConvFn(JGFunction):
    """y = conv(x, W / sigma) + bias with torch.nn.utils.spectral_norm semantics: one power iteration per TRAINING forward (u, v
    updated in place), sigma = u . W v; the gradient reaches W through 1 / sigma as well, with u and v held constant.
    Every forward owns the 16-bit copies of ITS W / sigma (a discriminator runs twice per step -- real and fake -- before one
    backward, each time with a new sigma)."""

    @staticmethod
    def forward(ctx, x, weight_orig, bias, mod):
        L = _lib.lib()
        m = mod.meta
        x = x.contiguous()
        B, H, W_, Cin = x.shape
        assert Cin == m.Cin, (Cin, m.Cin)
        dev = x.device
        K = m.R * m.S * m.Cin_real
        prepared = getattr(mod, "_sn_pass", None)
        if prepared is not None:               # this layer's power iteration and working copies were made with all the others'
            mod._sn_pass = None
            sn, idx = prepared
            w16, w16T = sn.views(idx)
            Ho, Wo = m.out_hw(H, W_)
            y = torch.empty((B, Ho, Wo, m.Cout), device=dev, dtype=x.dtype)
            conv_nt(x, w16, y, B=B, H=H, W=W_, Cin=Cin, Cout=m.Cout, R=m.R, S=m.S, pad=m.pad, stride=m.stride, Ho=Ho, Wo=Wo, ldx=Cin,
                    ldw=m.R * m.S * Cin, ldy=m.Cout, bias=m.bias_pad if m.bias_pad is not None else bias)
            ctx.save_for_backward(x, w16T)
            ctx.mod, ctx.sn = mod, (sn, idx)
            return y
        ctx.sn = None
        sigma = torch.empty(1, device=dev, dtype=torch.float32)
        Wp = weight_orig.data_ptr()            # arena slice: physical [Cout][R][S][Cin] fp32
        if mod.training:
            ws = torch.empty(K + m.Cout_real + 2, device=dev, dtype=torch.float32)
            check(L.jg_spectral_power_iter(Wp, mod.weight_u.data_ptr(), mod.weight_v.data_ptr(), sigma.data_ptr(), ws.data_ptr(), m.Cout_real,
                                           m.R * m.S, m.Cin_real, 1e-12, _st()), "jg_spectral_power_iter")
        else:       # eval: sigma from the stored vectors (torch.nn.utils.spectral_norm does not iterate in eval mode)
            Wl = weight_orig.detach().reshape(m.Cout_real, -1)
            sigma.copy_(torch.dot(mod.weight_u, Wl @ mod.weight_v).reshape(1))
        w16 = torch.empty((m.Cout, m.R, m.S, m.Cin), device=dev, dtype=x.dtype)
        w16T = torch.empty((m.Cin, m.R, m.S, m.Cout), device=dev, dtype=x.dtype) if ctx.needs_input_grad[0] else None
        check(L.jg_spectral_weights(_dt(x), Wp, sigma.data_ptr(), w16.data_ptr(), _p(w16T), m.Cout_real, m.R * m.S, m.Cin_real, m.Cout, m.Cin,
                                    _st()), "jg_spectral_weights")
        Ho, Wo = m.out_hw(H, W_)
        y = torch.empty((B, Ho, Wo, m.Cout), device=dev, dtype=x.dtype)
        conv_nt(x, w16, y, B=B, H=H, W=W_, Cin=Cin, Cout=m.Cout, R=m.R, S=m.S, pad=m.pad, stride=m.stride, Ho=Ho, Wo=Wo, ldx=Cin,
                ldw=m.R * m.S * Cin, ldy=m.Cout, bias=m.bias_pad if m.bias_pad is not None else bias)
        ctx.save_for_backward(x, w16T, sigma, mod.weight_u.clone(), mod.weight_v.clone())
        ctx.mod = mod
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        if ctx.sn is not None:
            x, w16T = ctx.saved_tensors
            sigma = u = v = None
        else:
            x, w16T, sigma, u, v = ctx.saved_tensors
        mod = ctx.mod
        m = mod.meta
        L = _lib.lib()
        dy = dy.contiguous()
        B, H, W_, Cin = x.shape
        _, Ho, Wo, Cout = dy.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if m.stride != 1:     # stride-s input gradient = stride-1 convolution over the zero-dilated dy (ops.conv2d_dgrad)
                Hd, Wd = H + 2 * m.pad - m.R + 1, W_ + 2 * m.pad - m.S + 1
                dyd = ops.dilate2d(dy, Hd, Wd, m.stride)
                conv_nt(dyd, w16T, dx, B=B, H=Hd, W=Wd, Cin=Cout, Cout=Cin, R=m.R, S=m.S, pad=m.R - 1 - m.pad, stride=1, Ho=H, Wo=W_,
                        ldx=Cout, ldw=m.R * m.S * Cout, ldy=Cin)
            else:
                conv_nt(dy, w16T, dx, B=B, H=Ho, W=Wo, Cin=Cout, Cout=Cin, R=m.R, S=m.S, pad=m.R - 1 - m.pad, stride=1, Ho=H, Wo=W_,
                        ldx=Cout, ldw=m.R * m.S * Cout, ldy=Cin)
        if ctx.needs_input_grad[1]:
            wg = mod.weight_orig.grad
            if wg is None:
                raise RuntimeError("spectral conv weight has no arena-backed .grad")
            K = m.R * m.S * m.Cin_real
            dwsn = ctx.sn[0].dwsn(ctx.sn[1]) if ctx.sn is not None else torch.zeros((m.Cout_real, K), device=dy.device, dtype=torch.float32)
            ktot = m.R * m.S * Cin
            splitk = ops._wgrad_splitk(((Cout + 127) // 128) * ((ktot + 127) // 128), B * Ho * Wo)
            wgrad_tn(dy, x, dwsn, B=B, H=H, W=W_, Cin=Cin, Cout=Cout, R=m.R, S=m.S, pad=m.pad, stride=m.stride, Ho=Ho, Wo=Wo, lddy=Cout, ldx=Cin,
                     lddw=K, dbias=mod.bias.grad if mod.bias is not None else None, Cin_out=m.Cin_real, Cout_out=m.Cout_real, splitk=splitk,
                     defer=ctx.sn is not None)      # (the per-layer fix below reads dwsn at once)
            if ctx.sn is not None:             # the sigma-gradient fix of all layers runs once, when the last one has delivered its dWsn
                ctx.sn[0].delivered(ctx.sn[1])
                return dx, None, None, None
            ws = torch.empty(1, device=dy.device, dtype=torch.float32)
            check(L.jg_spectral_wgrad_fix(dwsn.data_ptr(), mod.weight_orig.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(), wg.data_ptr(),
                                          ws.data_ptr(), m.Cout_real, m.R * m.S, m.Cin_real, _st()), "jg_spectral_wgrad_fix")
        return dx, None, None, None


class SpectralConv2d(nn.Module, JGConvNd):
    """`spectral_norm(nn.Conv2d(...))` of blocks.py:11-13 with the state_dict layout of torch.nn.utils.spectral_norm: parameters
    `bias`, `weight_orig`; buffers `weight_u` [Cout], `weight_v` [Cin * k * k]."""

    jg_wname = "weight_orig"

    def __init__(self, cin, cout, k, stride, padding, bias=True):
        super().__init__()
        ref = nn.Conv2d(cin, cout, k, stride, padding, bias=bias)          # the reference's default initialisation
        if bias:
            self.bias = nn.Parameter(ref.bias.detach().clone())
        else:
            self.register_parameter("bias", None)
        self.weight_orig = nn.Parameter(ref.weight.detach().clone())
        u = nn.functional.normalize(torch.randn(cout), dim=0, eps=1e-12)
        v = nn.functional.normalize(torch.randn(cin * k * k), dim=0, eps=1e-12)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)
        self.needs_dgrad = True
        self.jg_padding, self.jg_stride = padding, stride

    def forward(self, x):
        if self.meta is None:
            raise RuntimeError("SpectralConv2d used before ParamArena finalisation")
        return _SpectralConvFn.apply(x, self.weight_orig, self.bias, self)


class DownBlock(nn.Module):
    """blocks.py:182-200 (not separable): spectral conv 4x4 s2 p1 -> GroupNorm(c // 2, c) -> LeakyReLU(0.2)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.main = nn.Sequential(SpectralConv2d(cin, cout, 4, 2, 1), nn.GroupNorm(cout // 2, cout), nn.LeakyReLU(0.2, inplace=True))

    def forward(self, x):
        conv, gn = self.main[0], self.main[1]
        return ops.group_norm(conv(x), gn.num_groups, gn.weight, gn.bias, None, JG_ACT_LRELU, gn.eps)


CHANNEL_DICT = {4: 512, 8: 512, 16: 256, 32: 128, 64: 64, 128: 64, 256: 32, 512: 16, 1024: 8}


class SingleDisc(nn.Module):
    """discriminator.py:13-77 (head None, not separable, not patch)."""

    def __init__(self, nc, start_sz=256, end_sz=8):
        super().__init__()
        nfc = dict(CHANNEL_DICT)
        if start_sz not in nfc:          # sizes that are not powers of two: the closest entry (:36-39)
            start_sz = min(nfc, key=lambda s: abs(s - start_sz))
        self.start_sz = start_sz
        nfc[start_sz] = nc
        layers = []
        while start_sz > end_sz:
            layers.append(DownBlock(nfc[start_sz], nfc[start_sz // 2]))
            start_sz //= 2
        layers.append(SpectralConv2d(nfc[end_sz], 1, 4, 1, 0, bias=False))
        self.main = nn.Sequential(*layers)

    def forward(self, x):
        for layer in self.main:
            x = layer(x)
        return x                          # [B, h, w, 8] logits, channel 0 valid


class MultiScaleD(nn.Module):
    """discriminator.py:166-230 (conv = True, 4 discriminators, no conditioning): logits of the mini-discriminators flattened and
    concatenated, [B, sum h_i w_i]."""

    def __init__(self, channels, resolutions, num_discs=4):
        super().__init__()
        self.mini_discs = nn.ModuleDict({str(i): SingleDisc(nc=c, start_sz=r, end_sz=8)
                                         for i, (c, r) in enumerate(zip(channels[:num_discs], resolutions[:num_discs]))})

    def forward(self, features):
        outs = []
        for k, disc in self.mini_discs.items():
            lg = disc(features[k])
            outs.append(lg[..., 0].reshape(lg.shape[0], -1))
        return torch.cat(outs, dim=1)


# ---------------------------------------------------------------------------------------------------------------------
# frozen feature network
# ---------------------------------------------------------------------------------------------------------------------
class StandInEfficientNet(nn.Module):
    """torch stand-in for timm's `tf_efficientnet_lite0` with the attributes `_make_efficientnet` (projector.py:51-59) slices:
    conv_stem, bn1, blocks[0:2] -> stride 4 / 24 ch; blocks[2:3] -> stride 8 / 40; blocks[3:5] -> stride 16 / 112; blocks[5:9] ->
    stride 32 / 320.  Used AS IS (torch, CPU) by oracle/make_golden_projd.py to drive the unmodified reference."""

    def __init__(self):
        super().__init__()
        w = TF_EFFICIENTNET_LITE0_WIDTHS

        def stage(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, 4, 2, 1), nn.LeakyReLU(0.2))

        self.conv_stem = nn.Conv2d(3, 16, 4, 2, 1)
        self.bn1 = nn.LeakyReLU(0.2)
        self.blocks = nn.Sequential(nn.Identity(), stage(16, w[0]), stage(w[0], w[1]), nn.Identity(), stage(w[1], w[2]), nn.Identity(),
                                    nn.Identity(), nn.Identity(), stage(w[2], w[3]))


class _Stage(nn.Module):
    """one `pretrained.layer<i>` Sequential of the stand-in on the HIP ops (same child indices as the reference's slices)"""

    def __init__(self, items):
        super().__init__()
        for i, it in enumerate(items):
            self.add_module(str(i), it)

    def forward(self, x):
        for mod in self.children():
            x = _run_plain(mod, x)
        return x


def _run_plain(mod, x):
    if isinstance(mod, JGConv2d):
        return mod(x)
    if isinstance(mod, nn.LeakyReLU):
        return ops.activation(x, JG_ACT_LRELU)
    if isinstance(mod, nn.Identity):
        return x
    if isinstance(mod, (nn.Sequential, _Stage)):
        for sub in mod.children():
            x = _run_plain(sub, x)
        return x
    raise NotImplementedError(type(mod))


def _hip_stage(cin, cout):
    return nn.Sequential(JGConv2d(cin, cout, 4, padding=1, stride=2), nn.LeakyReLU(0.2))


# ---------------------------------------------------------------------------------------------------------------------
# tf_efficientnet_lite0 (timm `_gen_efficientnet_lite`, channel / depth multiplier 1.0): the architecture of the reference's frozen
# feature network (projector.py:251-255 `timm.create_model("tf_efficientnet_lite0", pretrained=True)`), cut into stages the way
# `_make_efficientnet` (projector.py:51-59) cuts it.  timm is not installed here and its weights cannot be downloaded: the ARCHITECTURE
# below restates timm's published definition (efficientnet.py `_gen_efficientnet_lite`, _efficientnet_blocks.py `DepthwiseSeparableConv`
# / `InvertedResidual`, TF "SAME" padding, BatchNorm eps 1e-3, ReLU6, no squeeze-excite) with timm's attribute names, so that a real
# `tf_efficientnet_lite0` checkpoint (or a reference `<epoch>_net_D_B_projected_d.pth`) loads key for key; the WEIGHTS are random until one
# is loaded ("backbone parity unpinned: timm absent").
#   (block type, repeats, kernel, stride, expansion, output channels)
# ---------------------------------------------------------------------------------------------------------------------
LITE0_STEM = 32
LITE0_ARCH = (("ds", 1, 3, 1, 1, 16), ("ir", 2, 3, 2, 6, 24), ("ir", 2, 5, 2, 6, 40), ("ir", 3, 3, 2, 6, 80), ("ir", 3, 5, 1, 6, 112),
              ("ir", 4, 5, 2, 6, 192), ("ir", 1, 3, 1, 6, 320))
LITE0_BN_EPS = 1e-3


def tf_same_pad(size, k, stride):
    """TF 'SAME' padding of timm's Conv2dSame: (low, high) zero padding of one axis"""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


class _FrozenBN(nn.Module):
    """BatchNorm2d of a frozen network in eval mode = per-channel affine (scale, shift); same state_dict entries as nn.BatchNorm2d"""

    def __init__(self, c, eps=LITE0_BN_EPS):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps = eps
        self._key, self._affine = None, None

    def affine(self):
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version, self.weight.data_ptr())
        if key != self._key:
            with torch.no_grad():
                sc = self.weight.detach().float() / torch.sqrt(self.running_var.float() + self.eps)
                sh = self.bias.detach().float() - self.running_mean.float() * sc
                if self._affine is None or self._affine[0].shape != sc.shape or self._affine[0].device != sc.device:
                    self._affine = (sc.contiguous(), sh.contiguous())
                else:       # refreshed IN PLACE: a captured hipGraph (cut_model._d_half_from_graph) holds these addresses
                    self._affine[0].copy_(sc)
                    self._affine[1].copy_(sh)
            self._key = key
        return self._affine


class _PWConv(nn.Conv2d, JGConvNd):
    """bias-free point-wise / stem convolution (timm creates every EfficientNet convolution without a bias) on the MFMA kernels"""

    def __init__(self, cin, cout, k=1, stride=1):
        nn.Conv2d.__init__(self, cin, cout, k, stride=stride, padding=0, bias=False)
        self.needs_dgrad = True
        self.jg_padding, self.jg_stride = 0, stride

    def forward(self, x):
        if self.meta is None:
            raise RuntimeError("EfficientNet convolution used before ParamArena finalisation")
        return ops.conv2d(x, self.meta, None, 1.0)


class _DWWeight(nn.Module):
    """`conv_dw.weight` [C, 1, k, k] of timm; the kernels read a cached fp32 [k*k][C] copy"""

    def __init__(self, c, k, stride):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c, 1, k, k))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.k, self.stride = k, stride
        self._key, self._taps = None, None

    def taps(self):
        key = (self.weight._version, self.weight.data_ptr())
        if key != self._key:
            with torch.no_grad():
                w = self.weight.detach().float()
                t = w.reshape(w.shape[0], -1).t()
                if self._taps is None or self._taps.shape != t.shape or self._taps.device != t.device:
                    self._taps = t.contiguous()
                else:       # in place, see FrozenBN.affine
                    self._taps.copy_(t)
            self._key = key
        return self._taps


class _DWAffineActFn(JGFunction):
    """relu6(scale * dwconv_kxk(x) + shift), TF SAME padding; backward = input gradient only (frozen weights)"""

    @staticmethod
    def forward(ctx, x, taps, scale, shift, k, stride, act):
        x = x.contiguous()
        B, H, W, C = x.shape
        (pt, _), (pl, _) = tf_same_pad(H, k, stride), tf_same_pad(W, k, stride)
        Ho, Wo = -(-H // stride), -(-W // stride)
        y = torch.empty((B, Ho, Wo, C), device=x.device, dtype=x.dtype)
        check(_lib.lib().jg_dwconv_affine_act_fwd(_dt(x), x.data_ptr(), taps.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), B, H, W, C,
                                                  k, stride, pt, pl, Ho, Wo, act, _st()), "jg_dwconv_affine_act_fwd")
        ctx.geo = (B, H, W, C, k, stride, pt, pl, Ho, Wo, act)
        ctx.save_for_backward(y, taps, scale)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        y, taps, scale = ctx.saved_tensors
        B, H, W, C, k, stride, pt, pl, Ho, Wo, act = ctx.geo
        dy = dy.contiguous()
        dx = torch.empty((B, H, W, C), device=dy.device, dtype=dy.dtype)
        check(_lib.lib().jg_dwconv_affine_act_bwd(_dt(dy), dy.data_ptr(), y.data_ptr(), taps.data_ptr(), scale.data_ptr(), dx.data_ptr(), B, H, W, C,
                                                  k, stride, pt, pl, Ho, Wo, act, _st()), "jg_dwconv_affine_act_bwd")
        return dx, None, None, None, None, None, None


class _ChanAffineActFn(JGFunction):
    @staticmethod
    def forward(ctx, x, scale, shift, act):
        x = x.contiguous()
        C = x.shape[-1]
        y = torch.empty_like(x)
        check(_lib.lib().jg_chan_affine_act_fwd(_dt(x), x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), x.numel() // C, C, act, _st()),
              "jg_chan_affine_act_fwd")
        ctx.act = act
        ctx.save_for_backward(y, scale)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        dy = dy.contiguous()
        C = y.shape[-1]
        dx = torch.empty_like(y)
        check(_lib.lib().jg_chan_affine_act_bwd(_dt(dy), dy.data_ptr(), y.data_ptr(), scale.data_ptr(), dx.data_ptr(), y.numel() // C, C, ctx.act, _st()),
              "jg_chan_affine_act_bwd")
        return dx, None, None, None


def _bn_act(x, bn, relu6):
    sc, sh = bn.affine()
    return _ChanAffineActFn.apply(x, sc, sh, 1 if relu6 else 0)


def _dw_bn_act(x, dw, bn):
    sc, sh = bn.affine()
    return _DWAffineActFn.apply(x, dw.taps(), sc, sh, dw.k, dw.stride, 1)


class _ZeroPadFn(JGFunction):
    """zero padding (top, left, bottom, right) of an NHWC map = the adjoint of a window crop (jg_crop2d)"""

    @staticmethod
    def forward(ctx, x, pt, pl, pb, pr):
        x = x.contiguous()
        B, H, W, C = x.shape
        y = torch.empty((B, H + pt + pb, W + pl + pr, C), device=x.device, dtype=x.dtype)
        check(_lib.lib().jg_crop2d(_dt(x), x.data_ptr(), y.data_ptr(), B, H + pt + pb, W + pl + pr, C, pt, pl, H, W, 1, _st()), "jg_crop2d")
        ctx.geo = (pt, pl, H, W)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        pt, pl, H, W = ctx.geo
        dy = dy.contiguous()
        B, Hp, Wp, C = dy.shape
        dx = torch.empty((B, H, W, C), device=dy.device, dtype=dy.dtype)
        check(_lib.lib().jg_crop2d(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, Hp, Wp, C, pt, pl, H, W, 0, _st()), "jg_crop2d")
        return dx, None, None, None, None


class DepthwiseSeparableConv(nn.Module):
    """timm _efficientnet_blocks.DepthwiseSeparableConv (no SE, pw_act False): dw kxk -> BN -> ReLU6 -> 1x1 -> BN (+ x)"""

    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.conv_dw, self.bn1 = _DWWeight(cin, k, stride), _FrozenBN(cin)
        self.conv_pw, self.bn2 = _PWConv(cin, cout), _FrozenBN(cout)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        h = _dw_bn_act(x, self.conv_dw, self.bn1)
        h = _bn_act(self.conv_pw(h), self.bn2, False)
        return _AddFn.apply(h, x) if self.has_skip else h


class InvertedResidual(nn.Module):
    """timm _efficientnet_blocks.InvertedResidual (no SE): 1x1 expand -> BN -> ReLU6 -> dw kxk -> BN -> ReLU6 -> 1x1 project -> BN (+ x)"""

    def __init__(self, cin, cout, k, stride, exp):
        super().__init__()
        mid = cin * exp
        self.conv_pw, self.bn1 = _PWConv(cin, mid), _FrozenBN(mid)
        self.conv_dw, self.bn2 = _DWWeight(mid, k, stride), _FrozenBN(mid)
        self.conv_pwl, self.bn3 = _PWConv(mid, cout), _FrozenBN(cout)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        h = _bn_act(self.conv_pw(x), self.bn1, True)
        h = _dw_bn_act(h, self.conv_dw, self.bn2)
        h = _bn_act(self.conv_pwl(h), self.bn3, False)
        return _AddFn.apply(h, x) if self.has_skip else h


def lite0_blocks():
    """the seven `blocks[i]` Sequentials of tf_efficientnet_lite0"""
    blocks, cin = [], LITE0_STEM
    for kind, rep, k, stride, exp, cout in LITE0_ARCH:
        stage = []
        for r in range(rep):
            s = stride if r == 0 else 1
            stage.append(DepthwiseSeparableConv(cin, cout, k, s) if kind == "ds" else InvertedResidual(cin, cout, k, s, exp))
            cin = cout
        blocks.append(nn.Sequential(*stage))
    return blocks


def _run_lite0(mod, x):
    if isinstance(mod, _PWConv) and mod.kernel_size[0] == 3:       # conv_stem: TF SAME padding made explicit
        (pt, pb), (pl, pr) = tf_same_pad(x.shape[1], 3, 2), tf_same_pad(x.shape[2], 3, 2)
        return mod(_ZeroPadFn.apply(x, pt, pl, pb, pr))
    if isinstance(mod, _FrozenBN):                                  # bn1 of the stem: BatchNormAct2d (BN + ReLU6)
        return _bn_act(x, mod, True)
    if isinstance(mod, (DepthwiseSeparableConv, InvertedResidual)):
        return mod(x)
    if isinstance(mod, (nn.Sequential, _Stage)):
        for sub in mod.children():
            x = _run_lite0(sub, x)
        return x
    raise NotImplementedError(type(mod))


class _Lite0Stage(nn.Module):
    """one `pretrained.layer<i>` of _make_efficientnet over the real architecture (same child indices as the reference's Sequentials)"""

    def __init__(self, items):
        super().__init__()
        for i, it in enumerate(items):
            self.add_module(str(i), it)

    def forward(self, x):
        for mod in self.children():
            x = _run_lite0(mod, x)
        return x


def make_efficientnet_lite0():
    """`_make_efficientnet(timm tf_efficientnet_lite0)` (projector.py:51-59): layer0 = conv_stem, bn1, blocks[0:2]; layer1 = blocks[2:3];
    layer2 = blocks[3:5]; layer3 = blocks[5:9]"""
    b = lite0_blocks()
    pre = nn.Module()
    pre.layer0 = _Lite0Stage([_PWConv(3, LITE0_STEM, 3, stride=2), _FrozenBN(LITE0_STEM), b[0], b[1]])
    pre.layer1 = _Lite0Stage([b[2]])
    pre.layer2 = _Lite0Stage([b[3], b[4]])
    pre.layer3 = _Lite0Stage([b[5], b[6]])
    return pre


class FeatureFusionBlockMatrix(nn.Module):
    """blocks.py:248-287: (x0 [+ x1]) -> bilinear x2 (align_corners=True) -> 1x1 conv to features // 2 when `expand`."""

    def __init__(self, features, expand=False):
        super().__init__()
        self.out_conv = JGConv2d(features, features // 2 if expand else features, 1)

    def forward(self, *xs):
        out = xs[0] if len(xs) == 1 else _AddFn.apply(xs[0], xs[1])
        out = bilinear(out, 2 * out.shape[1], 2 * out.shape[2], True)
        return self.out_conv(out)


class Proj(nn.Module):
    """projector.py:490-589 with proj_type 2: frozen backbone -> CCM (1x1 convs to cout * (1, 2, 4, 8)) -> CSM (top-down fusion)."""

    def __init__(self, cout=64, expand=True, interp=256, backbone="lite0"):
        super().__init__()
        w = TF_EFFICIENTNET_LITE0_WIDTHS
        if backbone == "lite0":        # the reference's architecture (timm tf_efficientnet_lite0), weights to be loaded
            pre = make_efficientnet_lite0()
        elif backbone == "standin":    # the 5-convolution stand-in that drives the reference fixture tests/golden/projd.pt
            pre = nn.Module()
            pre.layer0 = _Stage([JGConv2d(3, 16, 4, padding=1, stride=2), nn.LeakyReLU(0.2), nn.Identity(), _hip_stage(16, w[0])])
            pre.layer1 = _Stage([_hip_stage(w[0], w[1])])
            pre.layer2 = _Stage([nn.Identity(), _hip_stage(w[1], w[2])])
            pre.layer3 = _Stage([nn.Identity(), nn.Identity(), nn.Identity(), _hip_stage(w[2], w[3])])
        else:
            raise ValueError(f"projected-discriminator backbone {backbone!r}")
        self.pretrained = pre
        ccm = [cout, cout * 2, cout * 4, cout * 8] if expand else [cout] * 4
        sc = nn.Module()
        for i in range(4):
            setattr(sc, f"layer{i}_ccm", JGConv2d(w[i], ccm[i], 1))
        sc.layer3_csm = FeatureFusionBlockMatrix(ccm[3], expand=expand)
        sc.layer2_csm = FeatureFusionBlockMatrix(ccm[2], expand=expand)
        sc.layer1_csm = FeatureFusionBlockMatrix(ccm[1], expand=expand)
        sc.layer0_csm = FeatureFusionBlockMatrix(ccm[0])
        self.scratch = sc
        self.CHANNELS = [cout, cout, cout * 2, cout * 4] if expand else [cout] * 4
        self.RESOLUTIONS = [2 * (interp // s) for s in (4, 8, 16, 32)]       # the CSM doubles every map (projector.py:484-486)

    def forward(self, x):
        p, s = self.pretrained, self.scratch
        o0 = p.layer0(x)
        o1 = p.layer1(o0)
        o2 = p.layer2(o1)
        o3 = p.layer3(o2)
        c0, c1, c2, c3 = s.layer0_ccm(o0), s.layer1_ccm(o1), s.layer2_ccm(o2), s.layer3_ccm(o3)
        m3 = s.layer3_csm(c3)
        m2 = s.layer2_csm(m3, c2)
        m1 = s.layer1_csm(m2, c1)
        m0 = s.layer0_csm(m1, c0)
        return {"0": m0, "1": m1, "2": m2, "3": m3}


class ProjectedDiscriminator(nn.Module):
    """discriminator.py:233-286.  forward(x: [B, S, S, 8] 16-bit NHWC image, 3 valid channels) -> logits [B, N]."""

    def __init__(self, projector_model="efficientnet", interp=-1, img_size=256, cout=64, expand=True, backbone="lite0", pretrained_path=""):
        super().__init__()
        if projector_model not in ("efficientnet", "vitsmall"):
            raise NotImplementedError(f"D_proj_network_type={projector_model!r}: the convolutional ('efficientnet' = tf_efficientnet_lite0) and the ViT "
                                      "('vitsmall' = vit_small_patch16_224) projectors are built; vitbase / CLIP / SigLIP / DINOv2 / SegFormer / depth "
                                      "feature networks are not")
        self.interp = interp
        self.projector_model = projector_model
        size = interp if interp > 0 else img_size
        self.backbone_pretrained = False
        self.arena = None
        if projector_model == "vitsmall":         # projector.py:327-331 + discriminator.py:208-230 (modules/projected_d_vit.py)
            from .projected_d_vit import MultiScaleDVit, ProjVit

            self.backbone = "vit_small_patch16_224"
            self.freeze_feature_network = ProjVit(cout=cout, expand=expand, interp=size)
            self.freeze_feature_network.requires_grad_(False)
            if pretrained_path:
                self.load_pretrained_backbone(pretrained_path)
            else:
                import warnings

                warnings.warn("ProjectedDiscriminator: vit_small_patch16_224 is built with RANDOM frozen weights -- timm's pretrained checkpoint "
                              "cannot be downloaded here.  A projected GAN on random features is not the reference's discriminator: pass "
                              "jg_projd_pretrained=<vit_small_patch16_224 state_dict .pth> or load a reference D checkpoint.", stacklevel=2)
            self.discriminator = MultiScaleDVit(self.freeze_feature_network.CHANNELS, self.freeze_feature_network.RESOLUTIONS)
            self.per_sample = True          # LayerNorm backbone, Conv1d / MLP projector and heads: no batch statistics, no spectral norm (loss.py)
            return
        self.backbone = backbone
        self.freeze_feature_network = Proj(cout=cout, expand=expand, interp=size, backbone=backbone)
        self.freeze_feature_network.requires_grad_(False)
        if pretrained_path:
            self.load_pretrained_backbone(pretrained_path)
        elif backbone == "standin":
            import warnings

            warnings.warn("ProjectedDiscriminator: backbone='standin' is the 5-convolution fixture driver, NOT the reference's "
                          "tf_efficientnet_lite0 -- tests only (opt-in through jg_projd_backbone='standin')", stacklevel=2)
        else:
            import warnings

            warnings.warn("ProjectedDiscriminator: tf_efficientnet_lite0 is built with RANDOM frozen weights -- timm's pretrained checkpoint "
                          "cannot be downloaded here.  A projected GAN on random features is not the reference's discriminator: pass "
                          "jg_projd_pretrained=<tf_efficientnet_lite0 state_dict .pth> or load a reference D checkpoint.", stacklevel=2)
        self.discriminator = MultiScaleD(self.freeze_feature_network.CHANNELS, self.freeze_feature_network.RESOLUTIONS)
        self.arena = None

    def load_pretrained_backbone(self, path):
        """load a timm `tf_efficientnet_lite0` state_dict (keys `conv_stem.weight`, `bn1.*`, `blocks.<i>.<j>.*`) into the stages the way
        `_make_efficientnet` re-homes the modules; every backbone entry must be present"""
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        if self.projector_model == "vitsmall":       # timm VisionTransformer keys are the module's own
            res = self.freeze_feature_network.pretrained.load_state_dict(sd, strict=True)
            self.backbone_pretrained = True
            self._backbone_loaded()
            return res
        remap = {}
        for k, v in sd.items():
            if k.startswith("conv_stem."):
                remap["layer0.0." + k[len("conv_stem."):]] = v
            elif k.startswith("bn1."):
                remap["layer0.1." + k[len("bn1."):]] = v
            elif k.startswith("blocks."):
                i, rest = k[len("blocks."):].split(".", 1)
                i = int(i)
                stage, idx = (("layer0", i + 2) if i < 2 else ("layer1", 0) if i == 2 else ("layer2", i - 3) if i < 5 else ("layer3", i - 5))
                remap[f"{stage}.{idx}.{rest}"] = v
        res = self.freeze_feature_network.pretrained.load_state_dict(remap, strict=True)
        self.backbone_pretrained = True
        self._backbone_loaded()
        return res

    def _backbone_loaded(self):
        """a load_state_dict on a SUB-module does not reach the top-level post hook of the arena: the frozen 16-bit working copies are stale now"""
        arena = getattr(self, "arena", None)
        if arena is not None:
            arena.dirty = True

    def check_loaded_backbone(self, incompatible, source=""):
        """after a non-strict load_state_dict of a discriminator checkpoint: a checkpoint that does not carry the frozen backbone
        (missing / unexpected `freeze_feature_network.pretrained.*` keys) must not pass silently -- the backbone would stay random"""
        pre = "freeze_feature_network.pretrained."
        missing = [k for k in incompatible.missing_keys if k.startswith(pre)]
        unexpected = [k for k in incompatible.unexpected_keys if k.startswith(pre)]
        if missing or unexpected:
            raise RuntimeError(f"projected discriminator checkpoint {source}: backbone keys do not match the {self.backbone} feature network "
                               f"(missing {missing[:3]}{'...' if len(missing) > 3 else ''}, unexpected {unexpected[:3]}"
                               f"{'...' if len(unexpected) > 3 else ''}): the frozen features would stay randomly initialised")
        self.backbone_pretrained = True

    def train(self, mode=True):
        self.freeze_feature_network.train(False)
        self.discriminator.train(mode)
        self.training = mode
        return self

    def jg_finalize(self, device, act_dtype):
        from ..arena import ParamArena

        if self.arena is None:
            self.act_dtype = act_dtype
            self.arena = ParamArena(self, device, act_dtype, priority=(), frozen_prefixes=("freeze_feature_network.",))
        return self.arena

    def forward(self, x):
        if self.arena is not None:
            self.arena.ensure_fresh()
        if self.interp > 0 and (x.shape[1] != self.interp or x.shape[2] != self.interp):
            x = bilinear(x, self.interp, self.interp, False)
        feats = self.freeze_feature_network(x)
        if SN_GROUPED and self.discriminator.training and self.arena is not None and self.projector_model == "efficientnet":
            self._sn_prepare(x.dtype)
        return self.discriminator(feats)

    def _sn_prepare(self, dtype):
        """one power iteration + the 16-bit working copies of ALL spectral-norm convolutions ahead of the heads (they depend on the
        weights only): every SpectralConv2d finds its share in `_sn_pass` and skips its own five launches"""
        grp = getattr(self, "_sn_group", None)
        if grp is None or grp.dtype != dtype or not grp.still_valid():
            mods = [m for m in self.discriminator.modules() if isinstance(m, SpectralConv2d)]
            grp = self._sn_group = _SnGroup(mods, dtype)
        last = getattr(self, "_sn_last", None)
        if last is not None:
            last.flush()                     # a previous pass whose backward ended early (never in the CUT step: every head feeds the loss)
        sn = _SnPass(grp)
        self._sn_last = sn
        for i, mod in enumerate(grp.mods):
            mod._sn_pass = (sn, i)

"""ViT feature network of the projected discriminator on the HIP ops: `D_proj_network_type = "vitsmall"`, what
examples/example_gan_mario2sonic.json (BASELINE configs[2]) selects.  Mirror of /root/reference/models/modules/projected_d/projector.py
(`create_timm_model` :252-253 = timm `vit_small_patch16_224` created at `D_proj_interp`; `configure_get_feats_vit_timm` :138-153: the token
sequences behind blocks 2 / 5 / 8 / 11; `_make_projector` :431-487 with `nn.Conv1d` cross-channel mixing and `FeatureFusionBlockVector`
cross-scale mixing, blocks.py:232-247,290-320) and discriminator.py:208-230 (`MultiScaleD(conv=False)`: four Flatten / Linear / ReLU heads on
[B, C * T]).

What is and is not here
  * timm is not installed and its checkpoint cannot be downloaded: the ARCHITECTURE restates timm's published definition of the model name
    (patch 16, width 384, depth 12, 6 heads of 64, MLP ratio 4, qkv bias, LayerNorm eps 1e-6, exact GELU, class token + learned position
    embedding at the creation size, pre-norm blocks; `norm` / `head` exist for the state_dict only) with timm's attribute names, so that a
    real `vit_small_patch16_224` state_dict or a reference `<epoch>_net_D_B_projected_d.pth` loads key for key; the weights are random
    until one is loaded ("backbone parity unpinned: timm absent") and a loud warning says so.  tests/golden/projd_vit*.pt come from the
    UNMODIFIED reference driven over oracle/vit_small_torch.py.
  * Token-major layout: the reference transposes every feature to [B, C, T] for its Conv1d(k = 1) layers; here tokens stay [B, T, C]
    (a Conv1d(k = 1) over [B, C, T] IS a linear layer over the rows of [B, T, C]) and only the Flatten in front of the heads needs the
    [B, C, T] order (one 16-bit transposition, jg_transpose2d).
  * The twelve blocks run as ONE autograd node (`_VitTokensFn`): the network is frozen, so its backward is input gradients only -- per
    block four GEMMs, the fused attention backward, GELU', and the two LayerNorm input gradients with the residual branch folded in
    (`jg_layernorm_bwd_res`); the discriminator update (inputs without gradient) runs it forward-only and keeps nothing.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .._autograd import JGFunction
from .. import _lib, ops
from .._lib import check
from ..ops import JG_ACT_RELU, _dt, _gemm_geom, _p, _st, conv_nt
from .layers import JGConv1d, JGConv2d, JGLinear
from .projected_d import _AddFn

VIT_SMALL = dict(width=384, depth=12, heads=6, patch=16, mlp_ratio=4, eps=1e-6, num_classes=1000)
VIT_TAPS = (2, 5, 8, 11)


class _Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = JGLinear(dim, dim * 3)
        self.proj = JGLinear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = JGLinear(dim, hidden)
        self.fc2 = JGLinear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, heads, hidden, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, hidden)


class _PatchEmbed(nn.Module):
    def __init__(self, width, patch):
        super().__init__()
        self.proj = JGConv2d(3, width, patch, padding=0, stride=patch)


class VitSmallPatch16(nn.Module):
    """timm `vit_small_patch16_224` created with `img_size` (state_dict keys of timm's VisionTransformer)."""

    def __init__(self, img_size, cfg=VIT_SMALL):
        super().__init__()
        if img_size % cfg["patch"]:
            raise ValueError(f"ViT feature network: image size {img_size} is not a multiple of the patch size {cfg['patch']} (set D_proj_interp)")
        w = cfg["width"]
        self.cfg, self.img_size, self.grid = cfg, img_size, img_size // cfg["patch"]
        self.patch_embed = _PatchEmbed(w, cfg["patch"])
        self.cls_token = nn.Parameter(torch.zeros(1, 1, w))
        self.pos_embed = nn.Parameter(torch.randn(1, self.grid * self.grid + 1, w) * 0.02)
        self.blocks = nn.Sequential(*[_Block(w, cfg["heads"], w * cfg["mlp_ratio"], cfg["eps"]) for _ in range(cfg["depth"])])
        self.norm = nn.LayerNorm(w, eps=cfg["eps"])                 # not on the feature path (projector.py:138-153), state_dict only
        self.head = nn.Linear(w, cfg["num_classes"])
        nn.init.normal_(self.cls_token, std=1e-6)

    def forward(self, x):
        """x: [B, S, S, 8] 16-bit NHWC image (3 valid channels), S = img_size -> the four token sequences [B, T, C]"""
        if x.shape[1] != self.img_size or x.shape[2] != self.img_size:
            raise ValueError(f"ViT feature network created for {self.img_size} x {self.img_size} inputs, got {tuple(x.shape)} "
                             "(timm's PatchEmbed asserts the same)")
        return _VitTokensFn.apply(x, self)


def _lin(x, m, res=None):
    """y = x W^T + b (+ res) on [B, T, Cin] -> [B, T, Cout]"""
    B, T, _ = x.shape
    r4 = None if res is None else res.view(B, 1, T, res.shape[-1])
    return ops.conv2d_forward(x.view(B, 1, T, x.shape[-1]), m, r4, 1.0).view(B, T, m.Cout)


def _lin_dgrad(dy, m):
    B, T, _ = dy.shape
    return ops.conv2d_dgrad(dy.view(B, 1, T, dy.shape[-1]), m, (B, 1, T, m.Cin)).view(B, T, m.Cin)


def _ln_fwd(x, ln, want_stats):
    y = torch.empty_like(x)
    C = x.shape[-1]
    R = x.numel() // C
    mr = torch.empty((R, 2), device=x.device, dtype=torch.float32) if want_stats else None
    check(_lib.lib().jg_layernorm_fwd(_dt(x), x.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), y.data_ptr(), _p(mr), R, C, ln.eps, _st()),
          "jg_layernorm_fwd")
    return y, mr


def _ln_bwd_res(x, dy, ln, mr, res):
    dx = torch.empty_like(x)
    C = x.shape[-1]
    check(_lib.lib().jg_layernorm_bwd_res(_dt(x), x.data_ptr(), dy.data_ptr(), ln.weight.data_ptr(), mr.data_ptr(), _p(res), dx.data_ptr(),
                                          x.numel() // C, C, _st()), "jg_layernorm_bwd_res")
    return dx


def vit_attention_fwd(qkv, heads):
    """timm Attention core on the packed projection qkv [B, T, 3C] (channel order [3][heads][head_dim]) -> (a [B, T, C], logsumexp)"""
    B, T, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    a = torch.empty((B, T, C), device=qkv.device, dtype=qkv.dtype)
    L = torch.empty((B * heads, T), device=qkv.device, dtype=torch.float32)
    es = qkv.element_size()
    check(_lib.lib().jg_vit_attention_fwd(_dt(qkv), qkv.data_ptr(), qkv.data_ptr() + C * es, qkv.data_ptr() + 2 * C * es, C3, hd, a.data_ptr(),
                                          L.data_ptr(), B, T, heads, hd, hd ** -0.5, _st()), "jg_vit_attention_fwd")
    return a, L


def vit_attention_bwd(qkv, a, L, da, heads):
    B, T, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    dqkv = torch.empty_like(qkv)
    Dq = torch.empty((B * heads, T), device=qkv.device, dtype=torch.float32)
    es = qkv.element_size()
    check(_lib.lib().jg_vit_attention_bwd(_dt(qkv), qkv.data_ptr(), qkv.data_ptr() + C * es, qkv.data_ptr() + 2 * C * es, C3, hd, a.data_ptr(),
                                          L.data_ptr(), da.data_ptr(), dqkv.data_ptr(), dqkv.data_ptr() + C * es, dqkv.data_ptr() + 2 * C * es, C3,
                                          Dq.data_ptr(), B, T, heads, hd, hd ** -0.5, _st()), "jg_vit_attention_bwd")
    return dqkv


class _VitAttnFn(JGFunction):
    """stand-alone autograd form of the attention core (kernel tests)"""

    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = qkv.contiguous()
        a, L = vit_attention_fwd(qkv, heads)
        ctx.save_for_backward(qkv, a, L)
        ctx.heads = heads
        return a

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, da):
        qkv, a, L = ctx.saved_tensors
        return vit_attention_bwd(qkv, a, L, da.contiguous(), ctx.heads), None


def vit_attention(qkv, heads):
    return _VitAttnFn.apply(qkv, heads)


def _gelu(x):
    y = torch.empty_like(x)
    check(_lib.lib().jg_gelu_fwd(_dt(x), x.data_ptr(), y.data_ptr(), x.numel(), _st()), "jg_gelu_fwd")
    return y


def _gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    check(_lib.lib().jg_gelu_bwd(_dt(x), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _st()), "jg_gelu_bwd")
    return dx


class _VitTokensFn(JGFunction):
    """`get_feats` of configure_get_feats_vit_timm over the frozen ViT as one autograd node; backward = input gradient only."""

    @staticmethod
    def forward(ctx, x, net):
        ops._require_cuda(x)
        x = x.contiguous()
        L = _lib.lib()
        B, S, _, _ = x.shape
        heads = net.cfg["heads"]
        keep = ctx.needs_input_grad[0]
        pm = net.patch_embed.proj.meta
        if pm is None:
            raise RuntimeError("ViT feature network used before ParamArena finalisation")
        pe = ops.conv2d_forward(x, pm)                                    # [B, g, g, C]
        N, C = net.grid * net.grid, pm.Cout
        t = torch.empty((B, N + 1, C), device=x.device, dtype=x.dtype)
        check(L.jg_vit_tokens_fwd(_dt(x), pe.data_ptr(), net.cls_token.data_ptr(), net.pos_embed.data_ptr(), t.data_ptr(), B, N, C, _st()),
              "jg_vit_tokens_fwd")
        del pe
        saved, outs = [], []
        for i, blk in enumerate(net.blocks):
            h1, mr1 = _ln_fwd(t, blk.norm1, keep)
            qkv = _lin(h1, blk.attn.qkv.meta)
            a, lse = vit_attention_fwd(qkv, heads)
            t2 = _lin(a, blk.attn.proj.meta, res=t)
            h2, mr2 = _ln_fwd(t2, blk.norm2, keep)
            u = _lin(h2, blk.mlp.fc1.meta)
            t3 = _lin(_gelu(u), blk.mlp.fc2.meta, res=t2)
            if keep:
                saved.append((t, mr1, qkv, a, lse, t2, mr2, u))
            t = t3
            if i in VIT_TAPS:
                outs.append(t)
        ctx.net, ctx.saved, ctx.geo = net, saved, (B, S, N, C)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *douts):
        net, saved = ctx.net, ctx.saved
        ctx.saved = None
        B, S, N, C = ctx.geo
        heads = net.cfg["heads"]
        L = _lib.lib()
        tap = {i: (None if d is None else d.contiguous()) for i, d in zip(VIT_TAPS, douts)}
        dt = None
        for i in reversed(range(len(net.blocks))):
            if i in tap and tap[i] is not None:
                dt = tap[i] if dt is None else ops.axpby(dt, 1.0, tap[i], 1.0)
            if dt is None:
                continue
            blk = net.blocks[i]
            t, mr1, qkv, a, lse, t2, mr2, u = saved[i]
            saved[i] = None
            dg = _lin_dgrad(dt, blk.mlp.fc2.meta)
            du = _gelu_bwd(u, dg)
            dh2 = _lin_dgrad(du, blk.mlp.fc1.meta)
            dt2 = _ln_bwd_res(t2, dh2, blk.norm2, mr2, dt)               # dt2 = dt + LN2'(dh2)
            da = _lin_dgrad(dt2, blk.attn.proj.meta)
            dqkv = vit_attention_bwd(qkv, a, lse, da, heads)
            dh1 = _lin_dgrad(dqkv, blk.attn.qkv.meta)
            dt = _ln_bwd_res(t, dh1, blk.norm1, mr1, dt2)
        if dt is None:
            return None, None
        dpe = torch.empty((B, N, C), device=dt.device, dtype=dt.dtype)
        check(L.jg_vit_tokens_bwd(_dt(dt), dt.data_ptr(), dpe.data_ptr(), B, N, C, _st()), "jg_vit_tokens_bwd")
        pm = net.patch_embed.proj.meta
        P = pm.R
        K = pm.Cin * P * P
        dcol = torch.empty((B * N, K), device=dt.device, dtype=dt.dtype)
        # input gradient of the non-overlapping patch convolution = one GEMM with the flipped / transposed working weights [Cin*P*P, C],
        # then the adjoint of the patch gather
        conv_nt(dpe.view(B * N, C), pm.w16T.view(K, C), dcol, **_gemm_geom(B * N, K, C), ldx=C, ldw=C, ldy=K)
        dx = torch.empty((B, S, S, pm.Cin), device=dt.device, dtype=dt.dtype)
        check(L.jg_unpatchify(_dt(dt), dcol.data_ptr(), dx.data_ptr(), B, net.grid, net.grid, P, _st()), "jg_unpatchify")
        return dx, None


# ---------------------------------------------------------------------------------------------------------------------
# projector (CCM / CSM over token sequences) and the MLP heads
# ---------------------------------------------------------------------------------------------------------------------
class FeatureFusionBlockVector(nn.Module):
    """blocks.py:232-247,290-320: (x0 [+ x1]) -> Conv1d(k = 1) to features // 2 when `expand` (no resize for token sequences)."""

    def __init__(self, features, expand=False):
        super().__init__()
        self.out_conv = JGConv1d(features, features // 2 if expand else features, 1)

    def forward(self, *xs):
        out = xs[0] if len(xs) == 1 else _AddFn.apply(xs[0], xs[1])
        return self.out_conv(out)


class ProjVit(nn.Module):
    """projector.py:490-589 with a ViT feature network, proj_type 2: tokens behind blocks 2 / 5 / 8 / 11 -> Conv1d CCM to
    cout * (1, 2, 4, 8) -> top-down FeatureFusionBlockVector CSM.  CHANNELS / RESOLUTIONS as `_make_projector` computes them
    (RESOLUTIONS = the token count: projector.py:444-446)."""

    def __init__(self, cout=64, expand=True, interp=256, cfg=VIT_SMALL):
        super().__init__()
        self.pretrained = VitSmallPatch16(interp, cfg)
        w = cfg["width"]
        ccm = [cout, cout * 2, cout * 4, cout * 8] if expand else [cout] * 4
        sc = nn.Module()
        for i in range(4):
            setattr(sc, f"layer{i}_ccm", JGConv1d(w, ccm[i], 1))
        sc.layer3_csm = FeatureFusionBlockVector(ccm[3], expand=expand)
        sc.layer2_csm = FeatureFusionBlockVector(ccm[2], expand=expand)
        sc.layer1_csm = FeatureFusionBlockVector(ccm[1], expand=expand)
        sc.layer0_csm = FeatureFusionBlockVector(ccm[0])
        self.scratch = sc
        self.CHANNELS = [cout, cout, cout * 2, cout * 4] if expand else [cout] * 4
        self.RESOLUTIONS = [self.pretrained.grid ** 2 + 1] * 4

    def forward(self, x):
        s = self.scratch
        o0, o1, o2, o3 = self.pretrained(x)
        c0, c1, c2, c3 = s.layer0_ccm(o0), s.layer1_ccm(o1), s.layer2_ccm(o2), s.layer3_ccm(o3)
        m3 = s.layer3_csm(c3)
        m2 = s.layer2_csm(m3, c2)
        m1 = s.layer1_csm(m2, c1)
        m0 = s.layer0_csm(m1, c0)
        return {"0": m0, "1": m1, "2": m2, "3": m3}


class _Transpose2dFn(JGFunction):
    """[B, R, C] -> [B, C, R] (16-bit copy); its own adjoint"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, R, C = x.shape
        y = torch.empty((B, C, R), device=x.device, dtype=x.dtype)
        check(_lib.lib().jg_transpose2d(_dt(x), x.data_ptr(), y.data_ptr(), B, R, C, _st()), "jg_transpose2d")
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, C, R = dy.shape
        dx = torch.empty((B, R, C), device=dy.device, dtype=dy.dtype)
        check(_lib.lib().jg_transpose2d(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, C, R, _st()), "jg_transpose2d")
        return dx


class MultiScaleDVit(nn.Module):
    """discriminator.py:208-230 (conv = False): per level nn.Sequential(Flatten, Linear(C * T, 100), ReLU, Linear(100, 100), ReLU, Linear(100, 100))
    on the [B, C, T] feature; logits concatenated, [B, 400].  Same child indices (1 / 3 / 5) as the reference's Sequentials."""

    def __init__(self, channels, resolutions, num_discs=4):
        super().__init__()
        self.mini_discs = nn.ModuleDict()
        for i, (c, r) in enumerate(zip(channels[:num_discs], resolutions[:num_discs])):
            self.mini_discs[str(i)] = nn.Sequential(nn.Flatten(), JGLinear(c * r, 100), nn.ReLU(), JGLinear(100, 100), nn.ReLU(), JGLinear(100, 100))

    def forward(self, features):
        outs = []
        for k, mlp in self.mini_discs.items():
            f = features[k]                                   # [B, T, C] token-major
            B = f.shape[0]
            h = _Transpose2dFn.apply(f).view(1, B, -1)        # Flatten of [B, C, T]; the batch rows are the "pixels" of one GEMM
            h = ops.activation(mlp[1](h), JG_ACT_RELU)
            h = ops.activation(mlp[3](h), JG_ACT_RELU)
            h = mlp[5](h)
            outs.append(h.view(B, -1)[:, :100])
        return torch.cat(outs, dim=1)

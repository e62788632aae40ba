"""SegFormer attention generator (G_netG = segformer_attn_conv) on the HIP ops: mirror of
/root/reference/models/modules/segformer/ (`MixVisionTransformer` backbone.py:441-545, `PatchEmbed` :548-650,
`TransformerEncoderLayer` :333-421, `EfficientMultiheadAttention` :230-330, `MixFFN` :13-91, `SegformerHead` segformer_head.py:89-181,
`BaseDecodeHead.cls_seg` decode_head.py:181-186, `JoliSegformer` builder_from_scratch.py:8-44), `SegformerGenerator_attn`
(segformer_generator.py:93-164), `BaseGenerator_attn` (models/modules/attn_network.py:6-55) and the BatchNorm `ResnetDecoder` tail
(resnet_generator.py:274-347 with its default norm_layer).  Module tree and parameter names are the reference's, so `state_dict()`
keys / shapes match (`segformer.backbone.layers.0.1.0.attn.attn.in_proj_weight`, `final_conv.model.1.running_mean` ...).

Execution: a token sequence [B, N, C] is the NHWC map [B, H, W, C] (no nlc<->nchw transposes); 1x1 convs / linears / strided patch
and sr convs run on the MFMA implicit-GEMM kernels; LayerNorm, depth-wise 3x3 + GELU, the small-KV attention core, bilinear
resize+concat, BatchNorm, the attention composition and DropPath / Dropout2d scaling are csrc/segformer.hip.
Randomness (train mode): DropPath and Dropout2d draw uniforms from `rand_source(shape)` (default torch.rand on the device) in the
reference's order, so parity runs can inject the recorded draws."""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .._autograd import JGFunction
from .. import ops
from .. import ops_segformer as S
from ..ops import JG_ACT_RELU, JG_ACT_TANH
from .layers import JGConv2d, JGConvNd, JGConvTranspose2d, JGLinear

SEGFORMER_B0 = {
    "backbone": dict(in_channels=3, embed_dims=32, num_stages=4, num_layers=[2, 2, 2, 2], num_heads=[1, 2, 5, 8], patch_sizes=[7, 3, 3, 3],
                     sr_ratios=[8, 4, 2, 1], out_indices=[0, 1, 2, 3], mlp_ratio=4, qkv_bias=True, drop_rate=0, attn_drop_rate=0,
                     drop_path_rate=0.1),
    "decode_head": dict(in_channels=[32, 64, 160, 256], in_index=[0, 1, 2, 3], channels=256, dropout_ratio=0.1, num_classes=10,
                        align_corners=False),
}


class _Rand:
    """uniform source shared by the DropPath / Dropout2d sites of one generator"""

    def __init__(self):
        self.source = None      # callable(shape) -> uniform [0, 1) tensor on any device, or None for torch.rand on the activation's device

    def __call__(self, shape, device):
        if self.source is None:
            return torch.rand(shape, device=device)
        return self.source(shape).to(device=device, dtype=torch.float32).reshape(shape)

    def droppath_scales(self, probs, batch, device):
        """floor(keep + U) / keep for ALL DropPath sites of one backbone pass at once ([len(probs), batch]; three launches instead of four
        per site -- the CUT step runs 70 sites); None when the uniforms are injected (parity runs draw site by site, in the reference's
        order)"""
        if self.source is not None:
            return None
        keep = getattr(self, "_keep", None)
        if keep is None or keep.device != device or keep.shape[0] != len(probs):
            keep = self._keep = (1.0 - torch.tensor(probs, dtype=torch.float32)).to(device).view(-1, 1)
        return torch.rand((len(probs), batch), device=device).add_(keep).floor_().div_(keep)


FUSE_ADD_LN = os.environ.get("JG_FUSE_ADD_LN", "1") != "0"      # round 6: residual sum (+ DropPath scale) inside the LayerNorm pass that reads it


class _Pending:
    """identity + branch * scale[b], not yet formed"""

    __slots__ = ("identity", "branch", "scale")

    def __init__(self, identity, branch, scale):
        self.identity, self.branch, self.scale = identity, branch, scale

    def materialize(self):
        return _add(self.identity, self.branch) if self.scale is None else S.scale_add(self.branch, self.scale, self.identity)


def _ln_id(x, norm):
    """(residual input, LayerNorm(it)) for a tensor or a pending sum"""
    if isinstance(x, _Pending):
        return S.add_layer_norm_id(x.identity, x.branch, x.scale, norm.weight, norm.bias, 1e-6)
    return S.layer_norm_id(x, norm.weight, norm.bias, 1e-6)


class DropPath(nn.Module):
    """backbone.py:700-726; here fused with the residual add: identity + x * floor(keep + U) / keep."""

    def __init__(self, drop_prob, rand):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self._rand = [rand]

    def pending(self, identity, x):
        """(identity, branch, scale or None) of `add` WITHOUT the launch: the LayerNorm that consumes the sum forms it in its own pass
        (S.add_layer_norm_id, round 6); consumes this pass's scale row exactly as `add` does"""
        if self.drop_prob == 0.0 or not self.training:
            return _Pending(identity, x, None)
        pre = getattr(self, "_scale_row", None)
        if pre is not None:
            self._scale_row = None
            return _Pending(identity, x, pre)
        keep = 1.0 - self.drop_prob
        u = self._rand[0]((x.shape[0],), x.device)
        return _Pending(identity, x, (keep + u).floor() / keep)

    def add(self, identity, x):
        if self.drop_prob == 0.0 or not self.training:
            return _add(identity, x)
        pre = getattr(self, "_scale_row", None)      # this pass's row of _Rand.droppath_scales (set by MixVisionTransformer.compute_feat)
        if pre is not None:
            self._scale_row = None
            return S.scale_add(x, pre, identity)
        keep = 1.0 - self.drop_prob
        u = self._rand[0]((x.shape[0],), x.device)
        return S.scale_add(x, (keep + u).floor() / keep, identity)


class _AddFn(JGFunction):
    @staticmethod
    def forward(ctx, a, b):
        return ops.axpby(a, 1.0, b, 1.0)

    @staticmethod
    def backward(ctx, g):
        return g, g


def _add(a, b):
    return _AddFn.apply(a, b)


class JGMultiheadAttention(nn.Module, JGConvNd):
    """Parameters of nn.MultiheadAttention (packed `in_proj_weight` [3C, C], `in_proj_bias`, `out_proj`), batch-first execution."""

    jg_wname, jg_bname = "in_proj_weight", "in_proj_bias"

    def __init__(self, embed_dims, num_heads):
        super().__init__()
        assert embed_dims == 32 * num_heads, "the small-KV attention kernel is written for head dim 32 (all MiT variants)"
        self.embed_dim, self.num_heads = embed_dims, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dims, embed_dims))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dims))
        self.out_proj = JGLinear(embed_dims, embed_dims)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)
        self.needs_dgrad, self.jg_padding, self.jg_stride = True, 0, 1

    def forward(self, x_q, x_kv):
        C = self.embed_dim
        q = S.sliced_linear(x_q, self.meta, 0, C)
        kv = S.sliced_linear(x_kv, self.meta, C, 2 * C)
        return self.out_proj(S.attention_smallkv(q, kv, self.num_heads))


class EfficientMultiheadAttention(nn.Module):
    def __init__(self, embed_dims, num_heads, drop_path, sr_ratio, rand):
        super().__init__()
        self.attn = JGMultiheadAttention(embed_dims, num_heads)
        self.dropout_layer = DropPath(drop_path, rand)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = JGConv2d(embed_dims, embed_dims, sr_ratio, padding=0, stride=sr_ratio)
            self.norm = nn.LayerNorm(embed_dims, eps=1e-6)

    def forward(self, x, hw, identity, defer=False):
        B, N, C = x.shape
        kv = x
        if self.sr_ratio > 1:
            kv = self.sr(x.view(B, hw[0], hw[1], C))
            kv = S.layer_norm(kv.view(B, -1, C), self.norm.weight, self.norm.bias, self.norm.eps)
        if defer:
            return self.dropout_layer.pending(identity, self.attn(x, kv))
        return self.dropout_layer.add(identity, self.attn(x, kv))


class _PEConv(nn.Conv2d):
    """container of the depth-wise positional conv's parameters ([C, 1, 3, 3] + bias); runs fused with the GELU that follows it"""


class MixFFN(nn.Module):
    def __init__(self, embed_dims, feedforward_channels, drop_path, rand):
        super().__init__()
        fc1 = JGConv2d(embed_dims, feedforward_channels, 1)
        pe = _PEConv(feedforward_channels, feedforward_channels, 3, padding=1, groups=feedforward_channels)
        fc2 = JGConv2d(feedforward_channels, embed_dims, 1)
        self.layers = nn.Sequential(fc1, pe, nn.GELU(), nn.Dropout(0.0), fc2, nn.Dropout(0.0))
        self.dropout_layer = DropPath(drop_path, rand)

    def forward(self, x, hw, identity, defer=False):
        B, N, C = x.shape
        h = self.layers[0](x.view(B, hw[0], hw[1], C))
        h = S.dwconv3x3(h, self.layers[1].weight, self.layers[1].bias, gelu=True)
        h = self.layers[4](h)
        if defer:
            return self.dropout_layer.pending(identity, h.view(B, N, C))
        return self.dropout_layer.add(identity, h.view(B, N, C))


class TransformerEncoderLayer(nn.Module):
    def __init__(self, embed_dims, num_heads, feedforward_channels, drop_path_rate, sr_ratio, rand):
        super().__init__()
        self.norm1 = nn.LayerNorm(embed_dims, eps=1e-6)
        self.attn = EfficientMultiheadAttention(embed_dims, num_heads, drop_path_rate, sr_ratio, rand)
        self.norm2 = nn.LayerNorm(embed_dims, eps=1e-6)
        self.ffn = MixFFN(embed_dims, feedforward_channels, drop_path_rate, rand)

    def forward(self, x, hw):
        # (identity, norm(x)) from one node: the identity path's gradient is added inside the LayerNorm backward pass.  Round 6: the two residual
        # sums of the block are not launched: each is handed on as a pending (identity, branch, DropPath scale) and formed inside the LayerNorm
        # pass that reads it -- norm2 here, norm1 of the next block or the stage's final norm for the block's output
        defer = FUSE_ADD_LN and not ops.TORCH_OPS_BOUNDARY
        xi, h = _ln_id(x, self.norm1)
        x = self.attn(h, hw, identity=xi, defer=defer)
        xi, h = _ln_id(x, self.norm2)
        return self.ffn(h, hw, identity=xi, defer=defer)


class PatchEmbed(nn.Module):
    def __init__(self, in_channels, embed_dims, kernel_size, stride):
        super().__init__()
        self.projection = JGConv2d(in_channels, embed_dims, kernel_size, padding=kernel_size // 2, stride=stride)
        self.norm = nn.LayerNorm(embed_dims, eps=1e-6)

    def forward(self, x):
        x = self.projection(x)
        B, H, W, C = x.shape
        return S.layer_norm(x.view(B, H * W, C), self.norm.weight, self.norm.bias, 1e-6), (H, W)


class MixVisionTransformer(nn.Module):
    def __init__(self, rand, in_channels=3, embed_dims=64, num_stages=4, num_layers=(3, 4, 6, 3), num_heads=(1, 2, 4, 8), patch_sizes=(7, 3, 3, 3),
                 strides=(4, 2, 2, 2), sr_ratios=(8, 4, 2, 1), out_indices=(0, 1, 2, 3), mlp_ratio=4, qkv_bias=True, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.0, **unused):
        super().__init__()
        if drop_rate or attn_drop_rate or not qkv_bias:
            raise NotImplementedError("drop_rate / attn_drop_rate > 0 and qkv_bias=False are outside the built path")
        self.out_indices = tuple(out_indices)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(num_layers))]
        cur = 0
        self.layers = nn.ModuleList()
        for i, n in enumerate(num_layers):
            dims = embed_dims * num_heads[i]
            pe = PatchEmbed(in_channels, dims, patch_sizes[i], strides[i])
            blocks = nn.ModuleList([TransformerEncoderLayer(dims, num_heads[i], mlp_ratio * dims, dpr[cur + j], sr_ratios[i], rand) for j in range(n)])
            self.layers.append(nn.ModuleList([pe, blocks, nn.LayerNorm(dims, eps=1e-6)]))
            in_channels = dims
            cur += n

    def compute_feat(self, x, extract_layer_ids=()):
        outs, feats = [], []
        if self.training:         # every DropPath scale of this pass from one draw (the sites consume their rows in execution order)
            sites = getattr(self, "_dp_sites", None)
            if sites is None:
                sites = self._dp_sites = [m for m in self.modules() if isinstance(m, DropPath) and m.drop_prob > 0.0]
            if sites:
                sc = sites[0]._rand[0].droppath_scales([m.drop_prob for m in sites], x.shape[0], x.device)
                if sc is not None:
                    for m, row in zip(sites, sc):
                        m._scale_row = row
        for i, layer in enumerate(self.layers):
            x, hw = layer[0](x)
            for block in layer[1]:
                x = block(x, hw)
            if isinstance(x, _Pending):         # the last block's output sum is formed inside the stage's final LayerNorm pass
                x = S.add_layer_norm_id(x.identity, x.branch, x.scale, layer[2].weight, layer[2].bias, 1e-6)[1]
            else:
                x = S.layer_norm(x, layer[2].weight, layer[2].bias, 1e-6)
            x = x.view(x.shape[0], hw[0], hw[1], x.shape[-1])
            if i in self.out_indices:
                outs.append(x)
            if i in extract_layer_ids:
                feats.append(x)
        return outs, feats


HEAD_COMMUTE = os.environ.get("JG_HEAD_COMMUTE", "1") != "0"     # SegformerHead: fusion convolution before the resize (below)
HEAD_DROPOUT_FUSED = os.environ.get("JG_HEAD_DROPOUT_FUSED", "1") != "0"      # round 6: the head's Dropout2d inside jg_resize_sum / jg_resize_sum_bwd


class SegformerHead(nn.Module):
    def __init__(self, rand, in_channels, in_index, channels, dropout_ratio=0.1, num_classes=10, align_corners=False, **unused):
        super().__init__()
        if align_corners:
            raise NotImplementedError("align_corners=True")
        self.in_index, self.dropout_ratio, self._rand = list(in_index), float(dropout_ratio), [rand]
        self.num_classes = num_classes
        self.conv_seg = JGConv2d(channels, num_classes, 1)         # BaseDecodeHead registers conv_seg / dropout first (key order)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.convs = nn.ModuleList([nn.Sequential(OrderedDict([("conv", JGConv2d(c, channels, 1)), ("activate", nn.ReLU(inplace=True))]))
                                    for c in in_channels])
        self.fusion_conv = nn.Sequential(OrderedDict([("conv", JGConv2d(channels * len(in_channels), channels, 1)),
                                                      ("activate", nn.ReLU(inplace=True))]))

    def forward(self, inputs):
        inputs = [inputs[i] for i in self.in_index]
        Ho, Wo = inputs[0].shape[1:3]
        outs = [ops.activation(self.convs[i].conv(x), JG_ACT_RELU) for i, x in enumerate(inputs)]
        fm = self.fusion_conv.conv.meta
        C = outs[0].shape[-1]
        if (HEAD_COMMUTE and not ops.TORCH_OPS_BOUNDARY and 1 < len(outs) <= 4 and fm is not None and fm.Cin == fm.Cin_real == C * len(outs)
                and C % 8 == 0 and all(o.shape[-1] == C for o in outs) and tuple(outs[0].shape[1:3]) == (Ho, Wo)):
            # Round 6: the 1x1 fusion convolution and the bilinear resize are both linear and act on different axes, so
            #   fusion(cat_i resize(o_i)) = sum_i resize(W_i o_i) + b,   W_i = the i-th block of input channels of the fusion weight:
            # each term is convolved on its OWN map (64^2, 32^2, 16^2, 8^2 tokens per image instead of four times 64^2) and the four C-channel
            # results are resized and summed in one pass -- no 4 C-channel concatenation (268 MB per direction at batch 32), a quarter of the
            # GEMM work.  Same function; the 16-bit rounding points move from the resized maps to the convolved ones.
            ts = [S.sliced_in_conv(o, fm, i * C, with_bias=(i == 0)) for i, o in enumerate(outs)]
            if self.dropout is not None and self.training and HEAD_DROPOUT_FUSED:
                # nn.Dropout2d as the channel factor of the same pass (and of its adjoint): no scale pass over the 64 x 64 x C map per direction
                u = self._rand[0]((ts[0].shape[0], ts[0].shape[-1]), ts[0].device)
                return self.conv_seg(S.resize_sum(ts[0], ts[1:], JG_ACT_RELU, (u >= self.dropout_ratio).float() / (1.0 - self.dropout_ratio)))
            h = S.resize_sum(ts[0], ts[1:], JG_ACT_RELU)
        else:
            h = ops.activation(self.fusion_conv.conv(S.resize_concat(outs, Ho, Wo)), JG_ACT_RELU)
        if self.dropout is not None and self.training:        # nn.Dropout2d: whole channels, scaled by 1 / (1 - p)
            u = self._rand[0]((h.shape[0], h.shape[-1]), h.device)
            h = S.scale_add(h, (u >= self.dropout_ratio).float() / (1.0 - self.dropout_ratio), None, per_channel=True)
        return self.conv_seg(h)


class JoliSegformer(nn.Module):
    def __init__(self, cfg, rand):
        super().__init__()
        self.backbone = MixVisionTransformer(rand, **cfg["backbone"])
        self.decode_head = SegformerHead(rand, **cfg["decode_head"])
        if "auxiliary_head" in cfg:
            self.auxiliary_head = SegformerHead(rand, **cfg["auxiliary_head"])

    def extract_feat(self, img, extract_layer_ids=()):
        return self.backbone.compute_feat(img, extract_layer_ids)


class ResnetDecoderBN(nn.Module):
    """ResnetDecoder with its default norm_layer = nn.BatchNorm2d (conv-transpose bias=False): segformer_generator.py:135-140."""

    def __init__(self, input_nc, output_nc, ngf=64):
        super().__init__()
        model = []
        for i in range(2):
            mult = 2 ** (2 - i)
            model += [JGConvTranspose2d(ngf * mult, ngf * mult // 2, 3, stride=2, padding=1, output_padding=1, bias=False),
                      nn.BatchNorm2d(ngf * mult // 2), nn.ReLU(True)]
        model += [nn.ReflectionPad2d(3), JGConv2d(ngf, output_nc, 7, padding=0), nn.Tanh()]
        self.model = nn.Sequential(*model)
        assert input_nc == ngf * 4

    def forward(self, x):
        m = self.model
        x = S.batch_norm(m[0](x), m[1], JG_ACT_RELU)
        x = S.batch_norm(m[3](x), m[4], JG_ACT_RELU)
        x = ops.reflect_pad2d(x, 3)
        if ops.head7_ok(x, m[7].meta):
            return ops.head_conv7(x, m[7].meta, JG_ACT_TANH)      # 7x7 onto 3 channels + Tanh, row-packed (round 6)
        return ops.activation(m[7](x), JG_ACT_TANH)


class SegformerGenerator_attn(nn.Module):
    """segformer_generator.py:93-164 + attn_network.py:6-55 with final_conv=True.  Inputs / outputs NHWC 16-bit, image channels padded to 8."""

    def __init__(self, jg_dir, G_config_segformer, input_nc, img_size, nb_mask_attn, nb_mask_input, final_conv=True, padding_type="reflect"):
        super().__init__()
        if not final_conv:
            raise NotImplementedError("segformer_attn without the ResnetDecoder tail is not built")
        cfg = None
        path = os.path.join(jg_dir or "", G_config_segformer or "")
        if G_config_segformer and os.path.isfile(path):
            with open(path) as f:
                cfg = json.load(f)
        if cfg is None:
            cfg = json.loads(json.dumps(SEGFORMER_B0))          # models/configs/segformer/segformer_config_b0.json
        self.nb_mask_attn, self.nb_mask_input, self.input_nc = nb_mask_attn, nb_mask_input, input_nc
        cfg["backbone"]["in_channels"] = input_nc
        cfg["auxiliary_head"] = dict(cfg["decode_head"])
        cfg["decode_head"]["num_classes"] = 256
        cfg["auxiliary_head"]["num_classes"] = nb_mask_attn
        self.rand = _Rand()
        self.segformer = JoliSegformer(cfg, self.rand)
        self.final_conv = ResnetDecoderBN(256, input_nc * (nb_mask_attn - nb_mask_input), ngf=64)
        self.arena = None

    def jg_finalize(self, device, act_dtype):
        from ..arena import ParamArena

        if self.arena is None:
            self.act_dtype = act_dtype
            self.arena = ParamArena(self, device, act_dtype, priority=())
        return self.arena

    def compute_feats(self, input, extract_layer_ids=()):
        return self.segformer.extract_feat(input, extract_layer_ids)

    def get_feats(self, input, extract_layer_ids):
        if self.arena is not None:
            self.arena.ensure_fresh()
        return self.compute_feats(input, extract_layer_ids)[1]

    def feat_channels(self, extract_layer_ids):
        dims = [m[2].normalized_shape[0] for m in self.segformer.backbone.layers]
        return [dims[i] for i in extract_layer_ids]

    def forward(self, input):
        if self.arena is not None:
            self.arena.ensure_fresh()
        outs, _ = self.compute_feats(input)
        image = self.final_conv(self.segformer.decode_head(outs))
        logits = self.segformer.auxiliary_head(outs)
        return S.attention_compose(image, logits, input, self.nb_mask_attn, self.nb_mask_attn - self.nb_mask_input, min(self.input_nc, 3))

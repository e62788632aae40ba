"""ResNet generator of the CUT / CycleGAN family on the HIP ops: mirror of
/root/reference/models/modules/resnet_architecture/resnet_generator.py (`ResnetBlock` :11-95, `ResnetGenerator` :98-164,
`ResnetEncoder` :167-271, `ResnetDecoder` :274-347) for InstanceNorm (affine=False, hence conv bias=True,
models/modules/utils.py:101-104), reflect padding, no dropout, no spectral norm.

Same `nn.Sequential` indices as the reference, so `state_dict()` keys (`encoder.model.1.weight`,
`encoder.model.10.conv_block.5.bias`, `decoder.model.3.weight` ...) and the NCE tap ids of
`get_feats(x, [0, 4, 8, 12, 16])` are the reference's.  Execution: NHWC 16-bit; InstanceNorm + ReLU is one fused
normalisation pass (GroupNorm kernels with groups == C); reflection padding is a gather kernel in front of a pad-0
convolution; the stride-2 transposed convolutions run as stride-1 MFMA convolutions over the zero-dilated input.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .._autograd import JGFunction
from .. import ops
from ..ops import JG_ACT_NONE, JG_ACT_RELU, JG_ACT_TANH
from .layers import JGConv2d, JGConvTranspose2d


FUSE_REFLECT = os.environ.get("JG_FUSE_REFLECT", "1") != "0"


def _run(seq, x, taps=None, feats=None, upto=None):
    """Walk a reference-shaped nn.Sequential on the HIP ops.  InstanceNorm2d followed by ReLU is fused into one pass
    (the ReLU index then returns the same tensor, which is also what an in-place nn.ReLU(True) leaves in the
    reference's InstanceNorm output)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.ReflectionPad2d):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if (FUSE_REFLECT and m.padding[0] == 1 and isinstance(nxt, JGConv2d) and not (taps and i in taps)
                    and ops.reflect_conv_ok(x, nxt.meta)):
                x = ops.reflect_conv2d(x, nxt.meta)       # pad + conv in one launch (mirrored halo)
                i += 1
            else:
                x = ops.reflect_pad2d(x, m.padding[0])
        elif (isinstance(m, JGConv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.Tanh) and not (taps and (i in taps or i + 1 in taps))
              and (upto is None or i + 1 <= upto) and ops.head7_ok(x, m.meta)):
            x = ops.head_conv7(x, m.meta, JG_ACT_TANH)      # 7x7 onto <= 4 channels + Tanh, row-packed (round 6)
            i += 1
        elif isinstance(m, (JGConv2d, JGConvTranspose2d, ResnetBlock)):
            x = m(x)
        elif isinstance(m, nn.InstanceNorm2d):
            # nn.ReLU(True) is IN PLACE in the reference: a feature tapped at the InstanceNorm index is the tensor the
            # ReLU then overwrites, i.e. the post-ReLU values -- exactly what the fused pass produces
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = ops.group_norm(x, x.shape[-1], None, None, None, JG_ACT_RELU if fuse else JG_ACT_NONE, m.eps)
            if fuse:
                if taps and i in taps:
                    feats.append(x)
                i += 1      # the ReLU is done
        elif isinstance(m, nn.ReLU):
            x = ops.activation(x, JG_ACT_RELU)
        elif isinstance(m, nn.Tanh):
            x = ops.activation(x, JG_ACT_TANH)
        else:
            raise NotImplementedError(type(m))
        if taps and i in taps:
            feats.append(x)
        if upto is not None and i >= upto:
            break
        i += 1
    return x


RES_ID = os.environ.get("JG_RES_ID", "1") != "0"
NORM_ADD = os.environ.get("JG_NORM_ADD", "1") != "0"


class ResnetBlock(nn.Module):
    """resnet_generator.py:11-95: x + [ReflPad1, Conv3, IN, ReLU, ReflPad1, Conv3, IN](x)."""

    def __init__(self, dim, padding_type="reflect", use_bias=True):
        super().__init__()
        if padding_type != "reflect":
            raise NotImplementedError("G_padding_type=%r (only 'reflect' is built)" % padding_type)
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), JGConv2d(dim, dim, 3, padding=0), nn.InstanceNorm2d(dim), nn.ReLU(True),
            nn.ReflectionPad2d(1), JGConv2d(dim, dim, 3, padding=0), nn.InstanceNorm2d(dim))
        for c in (self.conv_block[1], self.conv_block[5]):
            if not use_bias:
                c.bias = None

    def forward(self, x):
        cb = list(self.conv_block)
        if RES_ID and FUSE_REFLECT and not ops.TORCH_OPS_BOUNDARY and ops.reflect_conv_ok(x, cb[1].meta):
            # round 6: the first convolution hands x back as the identity of the sum, so that the residual branch's gradient is added inside
            # its input-gradient launch (9 blocks x 2 backward passes of a 67 MB accumulation kernel per CUT step)
            xi, h = ops.reflect_conv2d_id(x, cb[1].meta)
            h = _run(cb[2:-1], h)
        else:
            xi, h = x, _run(cb[:-1], x)
        if NORM_ADD and not ops.TORCH_OPS_BOUNDARY:       # ... and the sum itself is formed in the apply pass of the block's last InstanceNorm
            return ops.group_norm(h, h.shape[-1], None, None, None, JG_ACT_NONE, cb[-1].eps, add=xi)
        return _AddFn.apply(xi, _run(cb[-1:], h))     # out = x + conv_block(x)


class _AddFn(JGFunction):
    @staticmethod
    def forward(ctx, a, b):
        return ops.axpby(a, 1.0, b, 1.0)

    @staticmethod
    def backward(ctx, g):
        return g, g


class ResnetEncoder(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=64, n_blocks=6, padding_type="reflect"):
        super().__init__()
        model = [nn.ReflectionPad2d(3), JGConv2d(input_nc, ngf, 7, padding=0, needs_dgrad=True), nn.InstanceNorm2d(ngf), nn.ReLU(True)]
        for i in range(2):
            mult = 2 ** i
            model += [JGConv2d(ngf * mult, ngf * mult * 2, 3, padding=1, stride=2), nn.InstanceNorm2d(ngf * mult * 2), nn.ReLU(True)]
        for _ in range(n_blocks):
            model += [ResnetBlock(ngf * 4, padding_type)]
        self.model = nn.Sequential(*model)

    def compute_feats(self, input, extract_layer_ids=(), upto=None):
        feats = []
        feat = _run(self.model, input, taps=set(extract_layer_ids), feats=feats, upto=upto)
        return feat, feats

    def feat_channels(self, extract_layer_ids):
        """real channel count of each tapped activation (the reference reads feat.shape[1])."""
        c, out = self.model[1].in_channels, {}
        for i, m in enumerate(self.model):
            if isinstance(m, JGConv2d):
                c = m.out_channels
            if i in extract_layer_ids:
                out[i] = c
        return [out[i] for i in extract_layer_ids]

    def forward(self, input):
        return _run(self.model, input)


class ResnetDecoder(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=64, padding_type="reflect"):
        super().__init__()
        model = []
        for i in range(2):
            mult = 2 ** (2 - i)
            model += [JGConvTranspose2d(ngf * mult, ngf * mult // 2, 3, stride=2, padding=1, output_padding=1),
                      nn.InstanceNorm2d(ngf * mult // 2), nn.ReLU(True)]
        model += [nn.ReflectionPad2d(3), JGConv2d(ngf, output_nc, 7, padding=0), nn.Tanh()]
        self.model = nn.Sequential(*model)

    def forward(self, input):
        return _run(self.model, input)


class ResnetGenerator(nn.Module):
    """resnet_generator.py:98-164.  Inputs / outputs are NHWC 16-bit with the image channels zero-padded to 8."""

    def __init__(self, input_nc, output_nc, ngf=64, n_blocks=9, padding_type="reflect", use_dropout=False, use_spectral=False):
        super().__init__()
        if use_dropout or use_spectral:
            raise NotImplementedError("dropout / spectral norm in the ResNet generator are outside the built path")
        self.encoder = ResnetEncoder(input_nc, output_nc, ngf, n_blocks, padding_type)
        self.decoder = ResnetDecoder(input_nc, output_nc, ngf, padding_type)
        self.arena = None

    def jg_finalize(self, device, act_dtype):
        from ..arena import ParamArena

        if self.arena is None:
            self.act_dtype = act_dtype
            self.arena = ParamArena(self, device, act_dtype, priority=())
        return self.arena

    def compute_feats(self, input, extract_layer_ids=()):
        return self.encoder.compute_feats(input, extract_layer_ids)

    def get_feats(self, input, extract_layer_ids=()):
        """resnet_generator.py:151-153; the encoder stops after the last tapped layer (the rest never reaches the loss)."""
        if self.arena is not None:
            self.arena.ensure_fresh()
        return self.encoder.compute_feats(input, extract_layer_ids, upto=max(extract_layer_ids))[1]

    def feat_channels(self, extract_layer_ids):
        return self.encoder.feat_channels(list(extract_layer_ids))

    def forward(self, input):
        if self.arena is not None:
            self.arena.ensure_fresh()
        return self.decoder(self.encoder(input))

    # the encoder is a deterministic function of its input in training mode as well (InstanceNorm without running statistics, no dropout):
    # the features `get_feats(x)` returns are the activations `forward(x)` has already computed (cut_model `jg_nce_reuse_feats`)
    deterministic_encoder = True

    def forward_with_feats(self, input, extract_layer_ids):
        """(forward(input), get_feats(input, extract_layer_ids)) from ONE encoder pass"""
        if self.arena is not None:
            self.arena.ensure_fresh()
        feat, feats = self.encoder.compute_feats(input, extract_layer_ids)
        return self.decoder(feat), feats

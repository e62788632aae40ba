"""Attention ResNet generators (G_netG = resnet_attn / mobile_resnet_attn) on the HIP ops: mirror of
/root/reference/models/modules/resnet_architecture/resnet_generator.py (`resnet_block_attn` :350-385, `ResnetGenerator_attn` :388-557),
/root/reference/models/modules/mobile_modules.py (`SeparableConv2d` :4-40) and `BaseGenerator_attn`
(/root/reference/models/modules/attn_network.py:6-54), built by gan_networks.define_G :150-176.

Attribute names are the reference's (`conv1`, `conv1_norm`, ..., `resnet_blocks.<i>.conv1`, `deconv1_content`, `deconv3_attention`), so
`state_dict()` keys match; for the mobile variant a block's convolution is `conv1.conv.{0,2}` (depth-wise 3x3, reflect padded, + 1x1).

Execution: NHWC 16-bit.  InstanceNorm + ReLU is one fused normalisation pass; the reflect-padded 3x3 convolutions of the blocks run on
the halo-resident kernel (pad_mode 1, no padded copy); the depth-wise convolution of the mobile blocks runs as reflect-pad -> zero-padded
depth-wise kernel -> crop (the border outputs of the padded run are the only ones touched by the zero padding, and they are cut off);
the two stride-2 transposed convolutions of each head run over the zero-dilated input; the 10-way softmax + blend of the
(nb_mask_attn - nb_mask_input) generated images and the input is ONE kernel (`attention_compose`, shared with the SegFormer generator).

Built for G_padding_type = 'reflect', no spectral norm, no wavelet feature space (train_feat_wavelet).
"""
from __future__ import annotations

import os

import torch.nn as nn

from .. import ops
from .. import ops_segformer as S
from ..ops import JG_ACT_NONE, JG_ACT_RELU, JG_ACT_TANH
from .layers import JGConv2d, JGConvTranspose2d
from .resnet_generator import _AddFn


NORM_ADD = os.environ.get("JG_NORM_ADD", "1") != "0"      # round 6: the residual sum of a block inside the apply pass of its last InstanceNorm


def _in_act(x, norm, act, sums=None, add=None):
    return ops.group_norm(x, x.shape[-1], None, None, None, act, norm.eps, sums=sums, add=add)


def _conv_in_act(conv, x, norm, act, add=None):
    """InstanceNorm2d(conv(x)) + act with the statistics taken in the convolution's epilogue where that form exists (ops.conv2d_stats)"""
    if isinstance(conv, SeparableConv2d):
        y, sums = conv(x, want_stats=True)
    elif isinstance(conv, JGConv2d) and conv.meta is not None:
        y, sums = ops.conv2d_stats(x, conv.meta)
    else:
        y, sums = conv(x), None
    return _in_act(y, norm, act, sums, add)


def _reflect_conv3(x, conv):
    """nn.Conv2d(C, C, 3, 1, padding=1, padding_mode='reflect')"""
    if ops.reflect_conv_ok(x, conv.meta):
        return ops.reflect_conv2d(x, conv.meta)
    return conv(ops.reflect_pad2d(x, 1))


class _DWConv(nn.Conv2d):
    """container of the depth-wise 3x3 parameters ([C, 1, 3, 3] + bias), reflect padded"""


class SeparableConv2d(nn.Module):
    """mobile_modules.py:4-40 as used by resnet_block_attn: depth-wise 3x3 (reflect) -> InstanceNorm2d -> 1x1 convolution."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, padding_mode="reflect"):
        super().__init__()
        if kernel_size != 3 or stride != 1 or padding != 1 or padding_mode != "reflect":
            raise NotImplementedError("SeparableConv2d is built as the 3x3 / stride 1 / reflect-padded block convolution")
        self.conv = nn.Sequential(_DWConv(in_channels, in_channels, 3, stride=1, padding=1, padding_mode="reflect", groups=in_channels),
                                  nn.InstanceNorm2d(in_channels), JGConv2d(in_channels, out_channels, 1))

    def forward(self, x, want_stats=False):
        B, H, W, C = x.shape
        dw = self.conv[0]
        h = S.dwconv3x3(x, dw.weight, dw.bias, gelu=False, reflect=True)      # mirrored taps inside the kernel: no padded copy, no crop
        h = _in_act(h, self.conv[1], JG_ACT_NONE)
        if want_stats:              # (y, statistics of y for the InstanceNorm the caller applies next)
            return ops.conv2d_stats(h, self.conv[2].meta)
        return self.conv[2](h)


class resnet_block_attn(nn.Module):
    """resnet_generator.py:350-385: x + IN(conv2(relu(IN(conv1(x)))))."""

    def __init__(self, channel, kernel=3, stride=1, padding_type="reflect", mobile=False):
        super().__init__()
        self.mobile = mobile
        if mobile:
            self.conv1 = SeparableConv2d(channel, channel, kernel, stride, 1, padding_type)
        else:
            self.conv1 = JGConv2d(channel, channel, kernel, padding=0)
        self.conv1_norm = nn.InstanceNorm2d(channel)
        if mobile:
            self.conv2 = SeparableConv2d(channel, channel, kernel, stride, 1, padding_type)
        else:
            self.conv2 = JGConv2d(channel, channel, kernel, padding=0)
        self.conv2_norm = nn.InstanceNorm2d(channel)
        if not mobile:      # weight_init(0, 0.02) of the generator's constructor (:447) only reaches plain nn.Conv2d children
            for c in (self.conv1, self.conv2):
                nn.init.normal_(c.weight, 0.0, 0.02)
                nn.init.zeros_(c.bias)

    def _conv(self, conv, x):
        return conv(x) if self.mobile else _reflect_conv3(x, conv)

    def forward(self, x):
        add = x if (NORM_ADD and not ops.TORCH_OPS_BOUNDARY) else None      # x + IN(conv2(.)) formed in that InstanceNorm's apply pass
        if self.mobile:             # the point-wise convolution's epilogue takes the statistics of the InstanceNorm behind it
            h = _conv_in_act(self.conv1, x, self.conv1_norm, JG_ACT_RELU)
            h = _conv_in_act(self.conv2, h, self.conv2_norm, JG_ACT_NONE, add)
        else:
            h = _in_act(self._conv(self.conv1, x), self.conv1_norm, JG_ACT_RELU)
            h = _in_act(self._conv(self.conv2, h), self.conv2_norm, JG_ACT_NONE, add=add)
        return h if add is not None else _AddFn.apply(x, h)


class ResnetGenerator_attn(nn.Module):
    """resnet_generator.py:388-557 + attn_network.py:6-54.  Inputs / outputs are NHWC 16-bit with the image channels zero-padded to 8."""

    def __init__(self, input_nc, output_nc, nb_mask_attn, nb_mask_input, ngf=64, n_blocks=9, use_spectral=False, size=128,
                 padding_type="reflect", mobile=False, twice_resnet_blocks=False, freq_space=False):
        super().__init__()
        if use_spectral or freq_space or padding_type != "reflect":
            raise NotImplementedError("resnet_attn is built for G_padding_type='reflect' without spectral norm / wavelet feature space")
        self.nb_mask_attn, self.nb_mask_input = nb_mask_attn, nb_mask_input
        self.input_nc, self.output_nc, self.ngf, self.nb = input_nc, output_nc, ngf, n_blocks
        self.twice_resnet_blocks = twice_resnet_blocks
        self.conv1 = JGConv2d(input_nc, ngf, 7, padding=0)
        self.conv1_norm = nn.InstanceNorm2d(ngf)
        self.conv2 = JGConv2d(ngf, ngf * 2, 3, padding=1, stride=2)
        self.conv2_norm = nn.InstanceNorm2d(ngf * 2)
        self.conv3 = JGConv2d(ngf * 2, ngf * 4, 3, padding=1, stride=2)
        self.conv3_norm = nn.InstanceNorm2d(ngf * 4)
        self.resnet_blocks = nn.Sequential(*[resnet_block_attn(ngf * 4, 3, 1, padding_type, mobile) for _ in range(n_blocks)])
        n_img = nb_mask_attn - nb_mask_input
        self.deconv1_content = JGConvTranspose2d(ngf * 4, ngf * 2, 3, stride=2, padding=1, output_padding=1)
        self.deconv1_norm_content = nn.InstanceNorm2d(ngf * 2)
        self.deconv2_content = JGConvTranspose2d(ngf * 2, ngf, 3, stride=2, padding=1, output_padding=1)
        self.deconv2_norm_content = nn.InstanceNorm2d(ngf)
        self.deconv3_content = JGConv2d(ngf, output_nc * n_img, 7, padding=0)      # `self.input_nc = output_nc  # hack` (:431)
        self.deconv1_attention = JGConvTranspose2d(ngf * 4, ngf * 2, 3, stride=2, padding=1, output_padding=1)
        self.deconv1_norm_attention = nn.InstanceNorm2d(ngf * 2)
        self.deconv2_attention = JGConvTranspose2d(ngf * 2, ngf, 3, stride=2, padding=1, output_padding=1)
        self.deconv2_norm_attention = nn.InstanceNorm2d(ngf)
        self.deconv3_attention = JGConv2d(ngf, nb_mask_attn, 1)
        self.tanh = nn.Tanh()
        self.arena = None

    def jg_finalize(self, device, act_dtype):
        from ..arena import ParamArena

        if self.arena is None:
            self.act_dtype = act_dtype
            self.arena = ParamArena(self, device, act_dtype, priority=())
        return self.arena

    # ---- encoder + blocks (:491-515) -------------------------------------------------------------------------------------
    def compute_feats(self, input, extract_layer_ids=(), upto=None):
        x = _in_act(self.conv1(ops.reflect_pad2d(input, 3)), self.conv1_norm, JG_ACT_RELU)
        x = _in_act(self.conv2(x), self.conv2_norm, JG_ACT_RELU)
        x = _in_act(self.conv3(x), self.conv3_norm, JG_ACT_RELU)
        feats = []
        for layer_id, layer in enumerate(self.resnet_blocks):
            x = layer(x)
            if layer_id in extract_layer_ids:
                feats.append(x)
            if upto is not None and layer_id >= upto:
                break
        return x, feats

    def tapped_layers(self, extract_layer_ids):
        """ids that produce a feature: block indices only (an id >= n_blocks, or the -1 the reference rewrites to n_blocks, taps nothing)"""
        return [i for i in range(self.nb) if i in extract_layer_ids]

    def get_feats(self, input, extract_layer_ids=()):
        if self.arena is not None:
            self.arena.ensure_fresh()
        taps = self.tapped_layers(extract_layer_ids)
        return self.compute_feats(input, taps, upto=max(taps) if taps else 0)[1]

    def feat_channels(self, extract_layer_ids):
        return [self.ngf * 4 for _ in self.tapped_layers(extract_layer_ids)]

    # ---- the two heads (:517-557) --------------------------------------------------------------------------------------------
    def compute_attention_content(self, feat):
        x = feat
        if self.twice_resnet_blocks:
            for layer in self.resnet_blocks:
                x = layer(x)
        c = _in_act(self.deconv1_content(x), self.deconv1_norm_content, JG_ACT_RELU)
        c = _in_act(self.deconv2_content(c), self.deconv2_norm_content, JG_ACT_RELU)
        image = ops.activation(self.deconv3_content(ops.reflect_pad2d(c, 3)), JG_ACT_TANH)
        a = _in_act(self.deconv1_attention(x), self.deconv1_norm_attention, JG_ACT_RELU)
        a = _in_act(self.deconv2_attention(a), self.deconv2_norm_attention, JG_ACT_RELU)
        logits = self.deconv3_attention(a)
        return logits, image

    deterministic_encoder = True       # InstanceNorm, no dropout: see ResnetGenerator.forward_with_feats

    def forward_with_feats(self, input, extract_layer_ids):
        if self.arena is not None:
            self.arena.ensure_fresh()
        feat, feats = self.compute_feats(input, self.tapped_layers(extract_layer_ids))
        logits, image = self.compute_attention_content(feat)
        return S.attention_compose(image, logits, input, self.nb_mask_attn, self.nb_mask_attn - self.nb_mask_input, min(self.output_nc, 3)), feats

    def forward(self, input):
        if self.arena is not None:
            self.arena.ensure_fresh()
        feat, _ = self.compute_feats(input)
        logits, image = self.compute_attention_content(feat)
        return S.attention_compose(image, logits, input, self.nb_mask_attn, self.nb_mask_attn - self.nb_mask_input, min(self.output_nc, 3))

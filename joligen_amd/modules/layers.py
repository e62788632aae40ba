"""Parameter containers with the reference's state_dict names whose forward runs the HIP ops."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class JGConvNd:
    """Marker mix-in: a conv layer whose 16-bit working weights are managed by ParamArena."""

    needs_dgrad = True
    meta = None
    jg_padding = 0
    jg_stride = 1
    jg_wname = "weight"     # attribute names of the parameters (nn.MultiheadAttention calls them in_proj_weight / in_proj_bias)
    jg_bname = "bias"


class JGConv2d(nn.Conv2d, JGConvNd):
    """nn.Conv2d parameters/initialisation (so checkpoints and default init match the reference);
    forward = implicit-GEMM MFMA kernel on NHWC 16-bit activations, with an optional fused
    residual `y = conv(x) + bias + res_scale * res`."""

    def __init__(self, cin, cout, k, padding=0, stride=1, needs_dgrad=True):
        super().__init__(cin, cout, k, stride=stride, padding=padding)
        self.needs_dgrad = needs_dgrad
        self.jg_padding, self.jg_stride = padding, stride

    def forward(self, x, res=None, res_scale=1.0):
        if self.meta is None:
            raise RuntimeError("JGConv2d used before ParamArena finalisation (call net.jg_finalize(device))")
        return ops.conv2d(x, self.meta, res, res_scale)


class JGConv1d(nn.Conv1d, JGConvNd):
    """nn.Conv1d(k=1) container (attention qkv / proj_out); runs as a 1x1 conv over [B, T, C]."""

    def __init__(self, cin, cout, k=1, needs_dgrad=True):
        assert k == 1
        super().__init__(cin, cout, 1)
        self.needs_dgrad = needs_dgrad
        self.jg_padding, self.jg_stride = 0, 1

    def forward(self, x, res=None, res_scale=1.0):
        if self.meta is None:
            raise RuntimeError("JGConv1d used before ParamArena finalisation")
        B, T, Cc = x.shape
        r4 = None if res is None else res.view(B, 1, T, res.shape[-1])
        y = ops.conv2d(x.view(B, 1, T, Cc), self.meta, r4, res_scale)
        return y.view(B, T, y.shape[-1])


class JGConvTranspose2d(nn.ConvTranspose2d, JGConvNd):
    """nn.ConvTranspose2d parameters (weight [Cin, Cout, k, k]); the arena keeps it as the stride-s CONVOLUTION whose
    input-gradient it is (O = Cin, I = Cout), so the forward reads the flipped / transposed 16-bit copy (`w16T`) over the
    zero-dilated input and the backward is that convolution's forward + weight gradient."""

    jg_transposed = True

    def __init__(self, cin, cout, k, stride=1, padding=0, output_padding=0, bias=True):
        super().__init__(cin, cout, k, stride=stride, padding=padding, output_padding=output_padding, bias=bias)
        if cin % 8 or cout % 8:
            raise NotImplementedError("transposed convolutions need channel counts that are multiples of 8")
        self.needs_dgrad = True
        self.jg_padding, self.jg_stride, self.jg_output_padding = padding, stride, output_padding

    def forward(self, x):
        if self.meta is None:
            raise RuntimeError("JGConvTranspose2d used before ParamArena finalisation")
        return ops.conv_transpose2d(x, self.meta, self.jg_output_padding)


class JGLinear(nn.Linear, JGConvNd):
    """nn.Linear parameters (2-D weight [out, in]) applied to 16-bit token sequences [B, N, in] as a 1x1 convolution on the MFMA
    implicit-GEMM kernel (the arena keeps the 16-bit working copies like for a conv)."""

    def __init__(self, cin, cout, bias=True, needs_dgrad=True):
        super().__init__(cin, cout, bias=bias)
        self.needs_dgrad = needs_dgrad
        self.jg_padding, self.jg_stride = 0, 1

    def forward(self, x, res=None, res_scale=1.0):
        if self.meta is None:
            raise RuntimeError("JGLinear used before ParamArena finalisation")
        shp = x.shape
        x4 = x.reshape(shp[0], 1, -1, shp[-1])
        r4 = None if res is None else res.reshape(shp[0], 1, -1, res.shape[-1])
        y = ops.conv2d(x4, self.meta, r4, res_scale)
        return y.view(*shp[:-1], y.shape[-1])


class GroupNorm(nn.Module):
    """Same attribute layout as the reference wrapper (unet_attn_utils.py:42-48): `.norm` is an
    nn.GroupNorm that only holds (weight, bias); the computation is ops.group_norm."""

    def __init__(self, group_size, channels):
        super().__init__()
        self.norm = nn.GroupNorm(group_size, channels)

    def forward(self, x, film=None, act=ops.JG_ACT_NONE):
        return ops.group_norm(x, self.norm.num_groups, self.norm.weight, self.norm.bias, film, act, self.norm.eps)


def zero_module(module):
    """unet_attn_utils.py:69-75."""
    for p in module.parameters():
        p.detach().zero_()
    return module

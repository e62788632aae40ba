"""PatchGAN discriminator on the HIP ops: mirror of /root/reference/models/modules/discriminators.py
(`NLayerDiscriminator` :10-118) for InstanceNorm (affine=False -> conv bias=True), no dropout / spectral norm / wavelets.
Same nn.Sequential indices as the reference (`model.0.weight` ... `model.11.bias`).  The 4x4 stride-2 convolutions run on the
generic MFMA implicit-GEMM kernel; their input gradients as stride-1 convolutions over the zero-dilated output gradient;
InstanceNorm + LeakyReLU(0.2) is one fused normalisation pass.  The 1-channel logit map is stored with 8 channels (7 zero)."""
from __future__ import annotations

import torch.nn as nn

from .. import ops
from ..ops import JG_ACT_LRELU, JG_ACT_NONE
from .layers import JGConv2d


class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, use_dropout=False, use_spectral=False, freq_space=False):
        super().__init__()
        if use_dropout or use_spectral or freq_space:
            raise NotImplementedError("dropout / spectral norm / wavelet input of the PatchGAN are outside the built path")
        kw, padw = 4, 1
        seq = [JGConv2d(input_nc, ndf, kw, padding=padw, stride=2), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [JGConv2d(ndf * nf_prev, ndf * nf_mult, kw, padding=padw, stride=2), nn.InstanceNorm2d(ndf * nf_mult),
                    nn.LeakyReLU(0.2, True)]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [JGConv2d(ndf * nf_prev, ndf * nf_mult, kw, padding=padw, stride=1), nn.InstanceNorm2d(ndf * nf_mult),
                nn.LeakyReLU(0.2, True)]
        seq += [JGConv2d(ndf * nf_mult, 1, kw, padding=padw, stride=1)]
        self.model = nn.Sequential(*seq)
        self.arena = None

    def jg_finalize(self, device, act_dtype):
        from ..arena import ParamArena

        if self.arena is None:
            self.act_dtype = act_dtype
            self.arena = ParamArena(self, device, act_dtype, priority=())
        return self.arena

    def forward(self, x):
        """x: [B,H,W,8] 16-bit NHWC (image channels zero-padded) -> logits [B,H',W',8] (channel 0 valid)."""
        if self.arena is not None:
            self.arena.ensure_fresh()
        mods = list(self.model)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, JGConv2d):
                x = m(x)
            elif isinstance(m, nn.InstanceNorm2d):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU)
                x = ops.group_norm(x, x.shape[-1], None, None, None, JG_ACT_LRELU if fuse else JG_ACT_NONE, m.eps)
                i += 1 if fuse else 0
            elif isinstance(m, nn.LeakyReLU):
                x = ops.activation(x, JG_ACT_LRELU)
            else:
                raise NotImplementedError(type(m))
            i += 1
        return x

"""DDPM UNet (attention in the mid block) on the HIP ops.

Mirror of /root/reference/models/modules/unet_generator_attn/unet_generator_attn.py
(`ResBlock` :143-266, `AttentionBlock` :269-319, `QKVAttentionLegacy` :322-347, `UNet` :390-695):
same constructor arguments, same module tree and therefore the same `state_dict()` keys and
shapes -- checkpoints interchange both ways.  What differs is the execution:

  * activations are NHWC 16-bit end to end (the reference: NCHW fp32);
  * GroupNorm + SiLU (+ FiLM scale-shift) is one fused op, the residual/skip add (and the
    1/sqrt(2) skip weight of the `efficient` variant) is fused into the second conv's epilogue;
  * every ResBlock's `emb_layers` projection runs as ONE stacked linear at the top of forward;
  * mid-block attention = InstanceNorm1d -> 1x1 conv -> MFMA QK^T -> fp32 softmax -> MFMA PV.

Not implemented (not on the SURVEY.md 8 path, raise at construction): wavelet `freq_space`,
`use_new_attention_order`, `use_checkpoint`, non-groupnorm norms.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from ..ops import JG_ACT_NONE, JG_ACT_SILU
from .layers import GroupNorm, JGConv1d, JGConv2d, zero_module


def normalization(channels, norm="groupnorm32"):
    """unet_attn_utils.py:94-113 (groupnorm / instancenorm / layernorm variants)."""
    if "groupnorm" in norm:
        return GroupNorm(int(norm.split("groupnorm")[1]), channels)
    if norm == "instancenorm":
        return GroupNorm(channels, channels)
    if norm == "layernorm":
        return GroupNorm(1, channels)
    raise NotImplementedError(f"norm {norm!r} is not implemented for the MI355X UNet")


class EmbedBlock(nn.Module):
    pass


class EmbedSequential(nn.Sequential, EmbedBlock):
    """unet_generator_attn.py:37-49; `emb` here is the per-block slice provider (see UNet.forward)."""

    def forward(self, x, emb):
        for layer in self:
            x = layer(x, emb) if isinstance(layer, EmbedBlock) else layer(x)
        return x


class ResBlock(EmbedBlock):
    def __init__(self, channels, emb_channels, dropout, norm, out_channel=None, use_conv=False,
                 use_scale_shift_norm=False, use_checkpoint=False, up=False, down=False, efficient=False,
                 freq_space=False):
        super().__init__()
        if freq_space or use_checkpoint or use_conv or not use_scale_shift_norm:
            raise NotImplementedError("ResBlock variant outside the SURVEY.md 8 hot path")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channel = out_channel or channels
        self.up, self.down, self.efficient = up, down, efficient
        self.updown = up or down
        self.in_layers = nn.Sequential(normalization(channels, norm), nn.SiLU(),
                                       JGConv2d(channels, self.out_channel, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * self.out_channel))
        self.out_layers = nn.Sequential(normalization(self.out_channel, norm), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(JGConv2d(self.out_channel, self.out_channel, 3, padding=1)))
        if self.out_channel == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = JGConv2d(channels, self.out_channel, 1)
        self.emb_slice = None  # (offset, n) into the stacked embedding projection, set by UNet

    def forward(self, x, emb):
        """x: [B,H,W,C] 16-bit.  emb: either the raw embedding [B, emb_channels] (fp32) or an
        `_EmbAll` holder with the pre-computed stacked projection."""
        if isinstance(emb, _EmbAll):
            off, n = self.emb_slice
            emb_out = emb.all[:, off:off + n]
        else:
            lin = self.emb_layers[1]
            emb_out = ops.linear(emb, lin.weight, lin.bias, JG_ACT_SILU)
        conv1 = self.in_layers[2]
        h = self.in_layers[0](x, act=JG_ACT_SILU)
        if self.updown:
            upd = ops.upsample_nearest2 if self.up else ops.avg_pool2
            if self.efficient and self.up:  # conv before the upsample (reference :239-242)
                h = upd(conv1(h))
                x = upd(x)
            else:
                h = upd(h)
                x = upd(x)
                h = conv1(h)
        else:
            h = conv1(h)
        h = self.out_layers[0](h, film=emb_out, act=JG_ACT_SILU)  # GN * (1+scale) + shift, SiLU
        if self.dropout and self.training:
            # nn.Dropout(p) of out_layers (unet_generator_attn.py:207-215 of the reference): element-wise Bernoulli(1 - p) mask scaled by
            # 1 / (1 - p), between the activation and the second convolution.  No BASELINE config sets G_dropout / a UNet dropout > 0:
            # the mask is a torch device op on the module-by-module graph (the fused schedule hands such networks over to it);
            # `dropout_rand` (callable(shape, device) -> uniforms) injects the draws for parity runs.
            src = getattr(self, "dropout_rand", None)
            u = src(h.shape, h.device) if src is not None else torch.rand(h.shape, device=h.device)
            keep = 1.0 - float(self.dropout)
            h = h * ((u < keep).to(h.dtype) * (1.0 / keep if keep > 0.0 else 0.0))
        skipw = 1.0 / math.sqrt(2) if self.efficient else 1.0
        if isinstance(self.skip_connection, nn.Identity):
            skip = x
        else:
            skip = self.skip_connection(x)
        return self.out_layers[3](h, res=skip, res_scale=skipw)  # skipw*skip + conv(h)


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False,
                 use_new_attention_order=False, use_transformer=False):
        super().__init__()
        if use_checkpoint or use_new_attention_order or use_transformer:
            raise NotImplementedError("AttentionBlock variant outside the SURVEY.md 8 hot path")
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0
            self.num_heads = channels // num_head_channels
        # normalization1d = InstanceNorm1d without affine -> no parameters (unet_attn_utils.py:116-117)
        self.qkv = JGConv1d(channels, channels * 3, 1)
        self.proj_out = zero_module(JGConv1d(channels, channels, 1))

    def forward(self, x):
        B, H, W, Cc = x.shape
        xf = x.view(B, H * W, Cc)
        xn = ops.group_norm(xf, Cc, None, None, None, JG_ACT_NONE, 1e-5)  # InstanceNorm1d over T
        qkv = self.qkv(xn)
        a = ops.attention_core(qkv, self.num_heads)
        return self.proj_out(a, res=xf, res_scale=1.0).view(B, H, W, Cc)


class _EmbAll:
    """Stacked output of every ResBlock's emb_layers: `all` is [B, sum(2*C_i)] fp32."""

    def __init__(self, all_):
        self.all = all_


class UNet(nn.Module):
    def __init__(self, image_size, in_channel, inner_channel, out_channel, res_blocks, attn_res, tanh,
                 n_timestep_train, n_timestep_test, norm, group_norm_size, cond_embed_dim, dropout=0,
                 channel_mults=(1, 2, 4, 8), conv_resample=True, use_checkpoint=False, use_fp16=False, num_heads=1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=True,
                 use_new_attention_order=False, efficient=False, freq_space=False):
        super().__init__()
        if freq_space or tanh or not resblock_updown or use_checkpoint:
            raise NotImplementedError("UNet variant outside the SURVEY.md 8 hot path")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channel, self.inner_channel, self.out_channel = image_size, in_channel, inner_channel, out_channel
        self.res_blocks, self.attn_res, self.dropout, self.channel_mults = res_blocks, attn_res, dropout, channel_mults
        self.num_heads, self.num_head_channels = num_heads, num_head_channels
        self.cond_embed_dim = cond_embed_dim
        self.efficient = efficient
        if norm == "groupnorm":
            norm = norm + str(group_norm_size)
        rb = dict(use_scale_shift_norm=use_scale_shift_norm, norm=norm, efficient=efficient)
        at = dict(num_head_channels=num_head_channels, use_new_attention_order=use_new_attention_order)

        ch = input_ch = int(channel_mults[0] * inner_channel)
        self.input_blocks = nn.ModuleList([EmbedSequential(JGConv2d(in_channel, ch, 3, padding=1))])
        input_block_chans = [ch]
        ds = 1
        for level, mult in enumerate(channel_mults):
            for _ in range(res_blocks[level]):
                layers = [ResBlock(ch, cond_embed_dim, 0.0, out_channel=int(mult * inner_channel), **rb)]
                ch = int(mult * inner_channel)
                if ds in attn_res:
                    layers.append(AttentionBlock(ch, num_heads=num_heads, **at))
                self.input_blocks.append(EmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mults) - 1:
                self.input_blocks.append(EmbedSequential(ResBlock(ch, cond_embed_dim, 0.0, out_channel=ch, down=True, **rb)))
                input_block_chans.append(ch)
                ds *= 2
        self.middle_block = EmbedSequential(
            ResBlock(ch, cond_embed_dim, dropout, **rb),
            AttentionBlock(ch, num_heads=num_heads, **at),
            ResBlock(ch, cond_embed_dim, dropout, **rb),
        )
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mults))[::-1]:
            for i in range(res_blocks[level] + 1):
                ich = input_block_chans.pop()
                layers = [ResBlock(ch + ich, cond_embed_dim, 0.0, out_channel=int(inner_channel * mult), **rb)]
                ch = int(inner_channel * mult)
                if ds in attn_res:
                    layers.append(AttentionBlock(ch, num_heads=num_heads_upsample, **at))
                if level and i == res_blocks[level]:
                    layers.append(ResBlock(ch, cond_embed_dim, 0.0, out_channel=ch, up=True, **rb))
                    ds //= 2
                self.output_blocks.append(EmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch, norm), nn.SiLU(),
                                 zero_module(JGConv2d(input_ch, out_channel, 3, padding=1)))
        self.beta_schedule = {
            "train": {"schedule": "linear", "n_timestep": n_timestep_train, "linear_start": 1e-6, "linear_end": 0.01},
            "test": {"schedule": "linear", "n_timestep": n_timestep_test, "linear_start": 1e-4, "linear_end": 0.09},
        }
        # slices of the stacked embedding projection, in named_parameters() order (= arena order)
        off = 0
        for m in self.modules():
            if isinstance(m, ResBlock):
                m.emb_slice = (off, 2 * m.out_channel)
                off += 2 * m.out_channel
        self.emb_total = off

    # -- forward ---------------------------------------------------------------------------
    def _emb_all(self, emb):
        arena = getattr(self, "_jg_arena_ref", None)
        if arena is None or ops.TORCH_OPS_BOUNDARY:
            return emb  # per-block projection
        W = arena.group_view("emb_layers.1.weight").view(self.emb_total, self.cond_embed_dim)
        b = arena.group_view("emb_layers.1.bias")
        dW = arena.group_view("emb_layers.1.weight", arena.g).view(self.emb_total, self.cond_embed_dim)
        db = arena.group_view("emb_layers.1.bias", arena.g)
        track = getattr(self, "_emb_track", None)
        if track is None:
            track = self._emb_track = [p for n, p in self.named_parameters() if ".emb_layers.1." in n]
        return _EmbAll(ops.linear_stacked(emb, W, b, dW, db, JG_ACT_SILU, track))

    def compute_feats(self, input, embed_gammas):
        if embed_gammas is None:
            embed_gammas = torch.ones((input.shape[0], self.cond_embed_dim), device=input.device)
        emb = self._emb_all(embed_gammas.float().contiguous())
        hs = []
        h = input
        for module in self.input_blocks:
            h = module(h, emb)
            hs.append(h)
        h = self.middle_block(h, emb)
        return h, hs, emb

    def forward(self, input, embed_gammas=None):
        """input: [B,H,W,Cpad] 16-bit NHWC (channels >= in_channel, zero padded to a multiple of 8)
        -> [B,H,W,8] (first out_channel channels valid).

        With a finalised arena the whole network runs as ONE autograd node on the fused schedule of
        unet_exec.py (same kernels, no concat / statistics / gradient-add passes); `jg_fused = False`
        selects the module-by-module graph below (what the parity tests compare it against)."""
        drop = bool(self.dropout) and self.training        # dropout > 0 in training: the module-by-module graph (ResBlock.forward) has the mask
        if getattr(self, "jg_fused", True) and getattr(self, "_jg_arena_ref", None) is not None and input.is_cuda and not drop \
                and not ops.TORCH_OPS_BOUNDARY:
            from .unet_exec import fused_unet

            if embed_gammas is None:
                embed_gammas = torch.ones((input.shape[0], self.cond_embed_dim), device=input.device)
            emb = self._emb_all(embed_gammas.float().contiguous())
            return fused_unet(self, input, emb.all)
        h, hs, emb = self.compute_feats(input, embed_gammas)
        for module in self.output_blocks:
            h = ops.cat_channels(h, hs.pop())
            h = module(h, emb)
        h = self.out[0](h, act=JG_ACT_SILU)
        return self.out[2](h)

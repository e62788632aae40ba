"""Patch feature sampler of the CUT family on the HIP ops: mirror of /root/reference/models/modules/cut_networks.py
`PatchSampleF` (:6-73): per tapped encoder layer, gather the SAME `num_patches` random spatial positions from every image of
the batch, run them through a 2-layer MLP (Linear - ReLU - Linear, created lazily from the feature widths) and L2-normalise.

Features arrive as NHWC 16-bit tensors [B, H, W, ld] (`channels[i]` real channels); everything downstream of the gather is
fp32: jg_gather_rows -> jg_linear (fp32 GEMM) x2 -> jg_l2norm.  state_dict keys are the reference's (`mlp_0.0.weight` ...)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..ops import JG_ACT_NONE, JG_ACT_RELU


class PatchSampleF(nn.Module):
    def __init__(self, use_mlp=False, init_type="normal", init_gain=0.02, nc=256):
        super().__init__()
        if init_type != "normal":
            raise NotImplementedError("only init_type='normal' (the reference default) is built")
        self.use_mlp, self.nc, self.mlp_init = use_mlp, nc, False
        self.init_type, self.init_gain = init_type, init_gain
        self.arena = None
        self.device = None

    def set_device(self, device):
        self.device = device

    def data_dependent_initialize(self, feats, channels=None):
        if self.use_mlp and not self.mlp_init:
            self.create_mlp(feats, channels)

    def create_mlp(self, feats, channels=None):
        """cut_networks.py:23-37; init_net 'normal': weights ~ N(0, init_gain), biases 0 (models/modules/utils.py:28-69)."""
        channels = channels or [f.shape[-1] for f in feats]
        for mlp_id, input_nc in enumerate(channels):
            mlp = nn.Sequential(nn.Linear(input_nc, self.nc), nn.ReLU(), nn.Linear(self.nc, self.nc))
            for m in mlp:
                if isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, 0.0, self.init_gain)
                    nn.init.constant_(m.bias, 0.0)
            setattr(self, "mlp_%d" % mlp_id, mlp)
        self.mlp_init = True

    def jg_finalize(self, device, act_dtype):
        from ..arena import ParamArena

        if self.arena is None:
            self.arena = ParamArena(self, device, act_dtype, priority=())
        return self.arena

    def draw_ids(self, feat, num_patches):
        """the `torch.randperm(H * W)[:P]` of forward() for one layer (see there)"""
        HW = feat.shape[1] * feat.shape[2]
        return torch.rand(HW, device=feat.device).topk(int(min(num_patches, HW))).indices

    def embed(self, x, feat_id):
        """the MLP + L2 normalisation of forward() on gathered rows [R, C] fp32 of layer `feat_id`"""
        if self.use_mlp:
            mlp = getattr(self, "mlp_%d" % feat_id)
            x = ops.linear(x, mlp[0].weight, mlp[0].bias, JG_ACT_NONE)
            x = ops.linear(x, mlp[2].weight, mlp[2].bias, JG_ACT_RELU)
        return ops.l2_normalize(x, 1e-7)

    def forward(self, feats, num_patches=64, patch_ids=None, channels=None):
        if num_patches <= 0:
            raise NotImplementedError("num_patches=0 (dense features) is outside the built path")
        return_ids, return_feats = [], []
        for feat_id, feat in enumerate(feats):
            B, H, W, ld = feat.shape
            C = ld if channels is None else channels[feat_id]
            if patch_ids is not None:
                patch_id = patch_ids[feat_id].reshape(-1)
            else:
                # `torch.randperm(H * W)[:P]` of the reference (cut_networks.py:57-60) as P distinct uniform positions from ONE Philox draw
                # + top-k: the same distribution, no host synchronisation, and capturable in a hipGraph (cut_model._g_capture) with the
                # SAME numbers as the eager launch sequence draws from the same generator state
                P = int(min(num_patches, H * W))
                patch_id = torch.rand(H * W, device=feat.device).topk(P).indices
            x = ops.gather_patches(feat, patch_id, C)                         # [B*P, C] fp32
            if self.use_mlp:
                mlp = getattr(self, "mlp_%d" % feat_id)
                x = ops.linear(x, mlp[0].weight, mlp[0].bias, JG_ACT_NONE)
                x = ops.linear(x, mlp[2].weight, mlp[2].bias, JG_ACT_RELU)    # ReLU applied on the GEMM's input read
            return_ids.append(patch_id.unsqueeze(0))
            return_feats.append(ops.l2_normalize(x, 1e-7))
        return return_feats, return_ids

"""Patch-contrastive criteria on the HIP kernels: /root/reference/models/modules/NCE/base_NCE.py (`BaseNCELoss` :6-77),
patchnce.py (`PatchNCELoss`), monce.py (`MoNCELoss` :12-33, 50 Sinkhorn iterations of sinkhorn.py:6-58, differentiated
through with respect to the query features).  Returns the per-patch loss vector like the reference."""
from __future__ import annotations

import torch.nn as nn

from ... import ops


class BaseNCELoss(nn.Module):
    monce = False

    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    def forward(self, feat_q, feat_k, current_batch, **unused_args):
        nimg = 1 if self.opt.alg_cut_nce_includes_all_negatives_from_minibatch else current_batch
        return ops.patch_nce_loss(feat_q, feat_k, nimg, self.opt.alg_cut_nce_T, self.opt.alg_cut_num_patches, self.monce)


class PatchNCELoss(BaseNCELoss):
    monce = False


class MoNCELoss(BaseNCELoss):
    monce = True

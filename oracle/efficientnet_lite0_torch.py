"""tf_efficientnet_lite0 as a plain torch module with timm's attribute names (TEST INFRASTRUCTURE ONLY).

The reference's projected discriminator builds its frozen feature network with `timm.create_model("tf_efficientnet_lite0", pretrained=True)`
(/root/reference/models/modules/projected_d/projector.py:251-255) and cuts it with `_make_efficientnet` (:51-59).  timm is an ABSENT
third-party dependency of the reference (requirements.txt: `timm`, version not pinned there) and its checkpoints cannot be downloaded here, so
this file restates timm's published architecture for that model name -- efficientnet.py `_gen_efficientnet_lite` (arch_def
ds_r1_k3_s1_e1_c16 / ir_r2_k3_s2_e6_c24 / ir_r2_k5_s2_e6_c40 / ir_r3_k3_s2_e6_c80 / ir_r3_k5_s1_e6_c112 / ir_r4_k5_s2_e6_c192 /
ir_r1_k3_s1_e6_c320, stem 32, ReLU6, no squeeze-excite, `tf_` = Conv2dSame padding + BatchNorm eps 1e-3) and _efficientnet_blocks.py
(`DepthwiseSeparableConv`, `InvertedResidual`, `BatchNormAct2d`) -- so that oracle/make_golden_projd.py can drive the UNMODIFIED reference
`ProjectedDiscriminator` over the real architecture (through a stubbed `timm.create_model`).  Parity of the backbone itself against timm is
UNPINNED (no timm to run); what the fixture pins is the product's HIP implementation of this architecture and everything downstream of it.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

ARCH = (("ds", 1, 3, 1, 1, 16), ("ir", 2, 3, 2, 6, 24), ("ir", 2, 5, 2, 6, 40), ("ir", 3, 3, 2, 6, 80), ("ir", 3, 5, 1, 6, 112),
        ("ir", 4, 5, 2, 6, 192), ("ir", 1, 3, 1, 6, 320))
STEM, BN_EPS = 32, 1e-3


def same_pad(size, k, stride):
    out = math.ceil(size / stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


class Conv2dSame(nn.Conv2d):
    """timm.layers.Conv2dSame: TensorFlow 'SAME' padding computed from the input size"""

    def forward(self, x):
        k, s = self.kernel_size[0], self.stride[0]
        (pt, pb), (pl, pr) = same_pad(x.shape[-2], k, s), same_pad(x.shape[-1], k, s)
        return F.conv2d(F.pad(x, (pl, pr, pt, pb)), self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


class BatchNormAct2d(nn.BatchNorm2d):
    """timm.layers.BatchNormAct2d: BatchNorm2d followed by the activation (state_dict of a plain BatchNorm2d)"""

    def __init__(self, c, act=True):
        super().__init__(c, eps=BN_EPS, momentum=0.01)
        self.apply_act = act

    def forward(self, x):
        x = super().forward(x)
        return F.relu6(x) if self.apply_act else x


class DepthwiseSeparableConv(nn.Module):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.conv_dw = Conv2dSame(cin, cin, k, stride, groups=cin, bias=False)
        self.bn1 = BatchNormAct2d(cin)
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn2 = BatchNormAct2d(cout, act=False)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        h = self.bn2(self.conv_pw(self.bn1(self.conv_dw(x))))
        return h + x if self.has_skip else h


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, k, stride, exp):
        super().__init__()
        mid = cin * exp
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = BatchNormAct2d(mid)
        self.conv_dw = Conv2dSame(mid, mid, k, stride, groups=mid, bias=False)
        self.bn2 = BatchNormAct2d(mid)
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = BatchNormAct2d(cout, act=False)
        self.has_skip = stride == 1 and cin == cout

    def forward(self, x):
        h = self.bn3(self.conv_pwl(self.bn2(self.conv_dw(self.bn1(self.conv_pw(x))))))
        return h + x if self.has_skip else h


class TfEfficientNetLite0(nn.Module):
    """the attributes `_make_efficientnet` reads: conv_stem, bn1, blocks (7 Sequentials)"""

    def __init__(self):
        super().__init__()
        self.conv_stem = Conv2dSame(3, STEM, 3, 2, bias=False)
        self.bn1 = BatchNormAct2d(STEM)
        blocks, cin = [], STEM
        for kind, rep, k, stride, exp, cout in ARCH:
            stage = []
            for r in range(rep):
                s = stride if r == 0 else 1
                stage.append(DepthwiseSeparableConv(cin, cout, k, s) if kind == "ds" else InvertedResidual(cin, cout, k, s, exp))
                cin = cout
            blocks.append(nn.Sequential(*stage))
        self.blocks = nn.Sequential(*blocks)

    def forward(self, x):
        return self.blocks(self.bn1(self.conv_stem(x)))

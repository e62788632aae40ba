"""Dev tool (GPU box): walk the encoder layer by layer keeping every intermediate gradient; compare two equivalent upstream
gradients (contiguous NHWC tensor vs the same values as a permuted NCHW tensor) and the CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch
import torch.nn as nn

import jg_oracle as O
from joligen_amd import ops
from joligen_amd.modules.resnet_generator import ResnetGenerator, _run

dtype = torch.float16
net = ResnetGenerator(3, 3, 16, n_blocks=2)
sd = {k: v.half().float() for k, v in O.synth_state_dict(net.state_dict(), 0).items()}
net.load_state_dict(sd)
net.jg_finalize(torch.device("cuda:0"), dtype)
net.arena.ensure_fresh()
g = torch.Generator().manual_seed(3)
x = (torch.rand(1, 3, 32, 32, generator=g) * 2 - 1).half().float()
R = torch.randn(1, 64, 8, 8, generator=g)
Rn = R.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()          # NHWC contiguous
mods = list(net.encoder.model)[:11]


def walk(up):
    net.arena.zero_grad()
    h = ops.to_nhwc(x.cuda(), dtype, 8)
    acts = []
    i = 0
    while i < len(mods):
        n = 1
        if isinstance(mods[i], nn.InstanceNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
            n = 2
        h = _run(nn.Sequential(*mods[i:i + n]), h)
        if h.requires_grad:
            h.retain_grad()
        acts.append((i, h))
        i += n
    h.backward(up)
    torch.cuda.synchronize()
    return [(i, a.grad.float().clone() if a.grad is not None else None) for i, a in acts], {k: p.grad.float().clone() for k, p in net.named_parameters()}


ga, pa = walk(Rn.clone())
gb, pb = walk(Rn.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1))
for (i, a), (_, b) in zip(ga, gb):
    if a is not None:
        print("act", i, "contig vs noncontig", float((a - b).norm() / (b.norm() + 1e-30)), float(a.norm()), float(b.norm()))
P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
f = O.resnet_encoder(P, x, 2, [10])[1][0]
(f * R).sum().backward()
for k in pa:
    if k.endswith("weight") and P[k].grad is not None:
        ref = P[k].grad
        print(k, "contig", float((pa[k].cpu() - ref).norm() / ref.norm()), "noncontig", float((pb[k].cpu() - ref).norm() / ref.norm()))

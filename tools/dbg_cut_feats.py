"""Dev tool (GPU box): gradient of sum(R . get_feats(x)[taps]) through the truncated encoder against the CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import jg_oracle as O
from joligen_amd import ops
from joligen_amd.modules.resnet_generator import ResnetGenerator

E = os.environ.get
ngf, nb, S, B = int(E("NGF", 16)), int(E("NB", 4)), int(E("S", 32)), int(E("B", 1))
taps = [int(i) for i in E("LAYERS", "13").split(",")]
dtype = torch.float16
net = ResnetGenerator(3, 3, ngf, n_blocks=nb)
sd = {k: v.half().float() for k, v in O.synth_state_dict(net.state_dict(), 0).items()}
net.load_state_dict(sd)
net.jg_finalize(torch.device("cuda:0"), dtype)
g = torch.Generator().manual_seed(3)
x = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).half().float()
P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
xo = x.clone().requires_grad_(True)
fo = O.resnet_encoder(P, xo, nb, taps)[1]
Rs = [torch.randn(f.shape, generator=g) for f in fo]
sum((f * r).sum() for f, r in zip(fo, Rs)).backward()
xd = ops.to_nhwc(x.cuda(), dtype, 8).requires_grad_(True)
fm = net.get_feats(xd, taps)
chans = net.feat_channels(taps)
for f, r, c in zip(fm, fo, chans):
    print("feat relerr", float((f.permute(0, 3, 1, 2)[:, :c].float().cpu() - r.detach()).norm() / r.detach().norm()))
loss = sum((f.float() * ops.to_nhwc(r.cuda(), dtype, f.shape[-1]).float()).sum() for f, r in zip(fm, Rs))
loss.backward()
torch.cuda.synchronize()
print("dx relerr", float((xd.grad.permute(0, 3, 1, 2)[:, :3].float().cpu() - xo.grad).norm() / xo.grad.norm()))
for k, p in net.named_parameters():
    ref = P[k].grad
    if ref is None or float(ref.norm()) < 1e-4:
        continue
    print("%.4f %-40s ref %.3e mine %.3e" % (float((p.grad.float().cpu() - ref).norm() / ref.norm()), k, float(ref.norm()), float(p.grad.float().norm())))

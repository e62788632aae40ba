"""Dev tool (GPU box): gradient through [encoder tap -> gather -> (MLP, l2norm)] against the CPU oracle, stage by stage."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import jg_oracle as O
from joligen_amd import ops
from joligen_amd.modules.cut_networks import PatchSampleF
from joligen_amd.modules.resnet_generator import ResnetGenerator

E = os.environ.get
ngf, nb, S, B, P = int(E("NGF", 16)), int(E("NB", 2)), int(E("S", 32)), int(E("B", 1)), int(E("P", 64))
taps = [int(i) for i in E("LAYERS", "10").split(",")]
stage = E("STAGE", "gather")
dtype = torch.float16
net = ResnetGenerator(3, 3, ngf, n_blocks=nb)
sd = {k: v.half().float() for k, v in O.synth_state_dict(net.state_dict(), 0).items()}
net.load_state_dict(sd)
net.jg_finalize(torch.device("cuda:0"), dtype)
g = torch.Generator().manual_seed(3)
x = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).half().float()
Pm = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
fo = O.resnet_encoder(Pm, x, nb, taps)[1]
chans = net.feat_channels(taps)
ids = [torch.randperm(f.shape[2] * f.shape[3], generator=g)[:P] for f in fo]
netF = PatchSampleF(use_mlp=True)
netF.data_dependent_initialize(None, chans)
sdF = O.synth_state_dict(netF.state_dict(), 3)
netF.load_state_dict(sdF)
netF.jg_finalize(torch.device("cuda:0"), dtype)


def poison():
    if not E("POISON"):
        return
    ts = [torch.full((n,), float("nan"), device="cuda", dtype=torch.float32) for n in (64, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 1 << 22, 1 << 24) for _ in range(8)]
    torch.cuda.synchronize()
    del ts


poison()
xd = ops.to_nhwc(x.cuda(), dtype, 8)
if E("XGRAD"):
    xd.requires_grad_(True)
LS = float(E("LS", "64"))
fm = net.get_feats(xd, taps)
for f in fm:
    f.retain_grad()
for f in fo:
    f.retain_grad()
if stage == "dense":
    outs_o = list(fo)
    outs_m = [f.permute(0, 3, 1, 2)[:, :c].float() for f, c in zip(fm, chans)]
elif stage == "perm":
    outs_o = [f.permute(0, 2, 3, 1).flatten(1, 2) for f in fo]
    outs_m = [f[..., :c].float().flatten(1, 2) for f, c in zip(fm, chans)]
elif stage == "torch":
    outs_o = [f.permute(0, 2, 3, 1).flatten(1, 2)[:, i, :].flatten(0, 1) for f, i in zip(fo, ids)]
    outs_m = [f[..., :c].float().flatten(1, 2)[:, i.cuda(), :].flatten(0, 1) for f, i, c in zip(fm, ids, chans)]
elif stage == "gather":
    outs_o = [f.permute(0, 2, 3, 1).flatten(1, 2)[:, i, :].flatten(0, 1) for f, i in zip(fo, ids)]
    outs_m = [ops.gather_patches(f, i.cuda(), c) for f, i, c in zip(fm, ids, chans)]
else:
    outs_o = O.patch_sample_f({k: v for k, v in sdF.items()}, fo, P, ids)
    outs_m = netF(fm, P, [i.cuda() for i in ids], chans)[0]
Rs = [torch.randn(o.shape, generator=g) for o in outs_o]
sum((o * r).sum() for o, r in zip(outs_o, Rs)).backward()
for a, b in zip(outs_m, outs_o):
    print("out relerr", float((a.float().cpu() - b.detach()).norm() / b.detach().norm()))
poison()
loss_m = sum((o * r.cuda()).sum() for o, r in zip(outs_m, Rs)) * LS
if E("EXPLICIT"):
    # same upstream gradient, handed to the engine directly at the tapped tensor
    (gf,) = torch.autograd.grad(loss_m, fm, retain_graph=True)
    print("explicit dfeat", tuple(gf.shape), gf.stride(), gf.dtype)
    gg = gf.contiguous() if E('EXPLICIT') == '2' else (gf.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1) if E('EXPLICIT') == '3' else gf.clone())
    print('passing', gg.stride(), gg.is_contiguous())
    torch.autograd.backward(fm, [gg])
else:
    loss_m.backward(create_graph=bool(E('CG')))
torch.cuda.synchronize()
for f, r in zip(fm, fo):
    c = r.shape[1]
    print('dfeat relerr', float((f.grad.permute(0, 3, 1, 2)[:, :c].float().cpu() / LS - r.grad).norm() / r.grad.norm()), tuple(f.grad.shape), f.grad.stride(), f.grad.is_contiguous())
for k, p in net.named_parameters():
    ref = Pm[k].grad
    if ref is None or float(ref.norm()) < 1e-4 or not k.endswith("weight"):
        continue
    print("%.4f %-40s ref %.3e mine %.3e" % (float((p.grad.float().cpu() / LS - ref).norm() / ref.norm()), k, float(ref.norm()), float(p.grad.float().norm()) / LS))

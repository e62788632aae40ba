"""Dev tool (GPU box): backward of the tapped encoder as a linear map -- several explicit upstream gradients in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch

import jg_oracle as O
from joligen_amd import ops
from joligen_amd.modules.resnet_generator import ResnetGenerator

dtype = torch.float16
tap = int(os.environ.get("LAYERS", "10"))
net = ResnetGenerator(3, 3, 16, n_blocks=2)
sd = {k: v.half().float() for k, v in O.synth_state_dict(net.state_dict(), 0).items()}
net.load_state_dict(sd)
net.jg_finalize(torch.device("cuda:0"), dtype)
g = torch.Generator().manual_seed(3)
x = (torch.rand(1, 3, 32, 32, generator=g) * 2 - 1).half().float()
Pm = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
fo = O.resnet_encoder(Pm, x, 2, [tap])[1][0]
keys = [k for k in Pm if k.endswith("weight") and k.startswith("encoder") and int(k.split(".")[2]) <= tap]
fm = net.get_feats(ops.to_nhwc(x.cuda(), dtype, 8), [tap])[0]
C, H, W = fo.shape[1:]
r = torch.randn(C * H * W, generator=g)
cases = {"as_chw": r.view(1, C, H, W), "as_hwc": r.view(1, H, W, C).permute(0, 3, 1, 2), "as_chw_again": r.view(1, C, H, W).clone(),
         "hwc_of_other": torch.randn(1, H, W, C, generator=g).permute(0, 3, 1, 2), "smooth": torch.ones(1, C, H, W) * torch.linspace(-1, 1, C).view(1, C, 1, 1)}
for name, G in cases.items():
    net.arena.zero_grad()
    torch.autograd.backward([fm], [G.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()], retain_graph=True)
    torch.cuda.synchronize()
    refs = torch.autograd.grad(fo, [Pm[k] for k in keys], G.contiguous(), retain_graph=True)
    P = dict(net.named_parameters())
    print(name, " ".join("%.3f" % float((P[k].grad.float().cpu() - ref).norm() / ref.norm()) for k, ref in zip(keys, refs)))
# the same upstream gradient produced by a torch loss expression on the [B, HW, C] view (what PatchSampleF's gather amounts to)
Rp = torch.randn(1, H * W, C, generator=g)
for name in ("loss_hwc", "loss_hwc_oracle_backward"):
    net.arena.zero_grad()
    loss = (fm[..., :C].float().flatten(1, 2) * Rp.cuda()).sum()
    loss.backward(retain_graph=True)
    torch.cuda.synchronize()
    if name == "loss_hwc":
        refs = torch.autograd.grad(fo, [Pm[k] for k in keys], Rp.view(1, H, W, C).permute(0, 3, 1, 2).contiguous(), retain_graph=True)
    else:
        for k in keys:
            Pm[k].grad = None
        (fo.permute(0, 2, 3, 1).flatten(1, 2) * Rp).sum().backward(retain_graph=True)
        refs = [Pm[k].grad for k in keys]
    P = dict(net.named_parameters())
    print(name, " ".join("%.3f" % float((P[k].grad.float().cpu() - ref).norm() / ref.norm()) for k, ref in zip(keys, refs)))

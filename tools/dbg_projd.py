"""dev: per-stage comparison of the HIP projected discriminator (lite0 backbone) with the CPU oracle on the projd_lite0.pt fixture inputs"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import torch.nn.functional as F

import jg_oracle as O
from joligen_amd import ops
from joligen_amd.modules.projected_d import ProjectedDiscriminator, bilinear

warnings.simplefilter("ignore")
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
g = torch.load(os.path.join(ROOT, "tests/golden/projd_lite0.pt"), weights_only=False)
P = O.synth_state_dict({k: torch.empty(g["shapes"][k]) for k in g["keys"]}, seed=5)
P = {k: (v.half().float() if (torch.is_floating_point(v) and not k.endswith(("weight_u", "weight_v"))) else v) for k, v in P.items()}
net = ProjectedDiscriminator(interp=256, img_size=64)
net.load_state_dict(P)
net.jg_finalize(torch.device("cuda:0"), dt)
net.train()
net.arena.ensure_fresh()
x = g["real"].half().float()
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-30))
xd = ops.to_nhwc(x.cuda(), dt, 8)
xi = bilinear(xd, 256, 256, False)
xr = F.interpolate(x, 256, mode="bilinear", align_corners=False)
print("interp", rel(xi.permute(0, 3, 1, 2)[:, :3].float(), xr))
pre = net.freeze_feature_network.pretrained
b = "freeze_feature_network.pretrained."
with torch.no_grad():
    outs_r = O.efficientnet_lite0_stages(P, xr, b)
    h = xi
    for i, layer in enumerate((pre.layer0, pre.layer1, pre.layer2, pre.layer3)):
        h = layer(h)
        print("stage", i, tuple(h.shape), rel(h.permute(0, 3, 1, 2).float(), outs_r[i]), float(outs_r[i].abs().mean()))
    feats = net.freeze_feature_network(xi)
    fr = O.projd_features(P, xr)
    for i in range(4):
        print("mixed", i, tuple(feats[str(i)].shape), rel(feats[str(i)].permute(0, 3, 1, 2).float(), fr[i]), float(fr[i].abs().mean()))
# block-by-block inside layer0 / layer1
with torch.no_grad():
    import types
    h = xi
    hr = xr
    from joligen_amd.modules.projected_d import _run_lite0
    mods = list(pre.layer0.children())
    h = _run_lite0(mods[0], h)
    hr = O._lite0_same_conv(hr, P[b + "layer0.0.weight"], 2)
    print("stem conv", rel(h.permute(0, 3, 1, 2).float(), hr))
    h = _run_lite0(mods[1], h)
    hr = O._lite0_bn(P, b + "layer0.1.", hr, True)
    print("stem bn", rel(h.permute(0, 3, 1, 2).float(), hr))

# gradient to the image through the frozen network, stage by stage: d(sum of stage output * fixed random r) / d input
for i in range(4):
    xq = xi.detach().clone().requires_grad_(True)
    h = xq
    for layer in (pre.layer0, pre.layer1, pre.layer2, pre.layer3)[: i + 1]:
        h = layer(h)
    gr = torch.Generator().manual_seed(i)
    r = torch.randn(outs_r[i].shape, generator=gr)
    rd = ops.to_nhwc(r.cuda(), dt, None) if r.shape[1] % 8 == 0 else None
    h.backward(rd * 64.0)
    xrq = xr.clone().requires_grad_(True)
    hr = O.efficientnet_lite0_stages(P, xrq, b)[i]
    (hr * r).sum().backward()
    print("dgrad through stage", i, rel(xq.grad.permute(0, 3, 1, 2)[:, :3].float() / 64.0, xrq.grad))

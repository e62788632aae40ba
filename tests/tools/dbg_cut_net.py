import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, jg_oracle as O
from joligen_amd import ops
from joligen_amd.modules.resnet_generator import ResnetGenerator
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-30))
def nchw(t, c): return t.permute(0, 3, 1, 2)[:, :c].float()
d = torch.device("cuda:0")
for name in ("small", "wide"):
    g = torch.load(f"tests/golden/cutnet_{name}.pt", weights_only=False); c, G = g["cfg"], g["G"]
    for dtype in (torch.float16, torch.bfloat16):
        net = ResnetGenerator(3, 3, c["ngf"], n_blocks=c["n_blocks"]); net.load_state_dict(O.synth_state_dict(net.state_dict(), seed=0)); net.jg_finalize(d, dtype)
        x = ops.to_nhwc(G["x"].to(d), dtype, 8).requires_grad_(True)
        out = net(x); out.backward(ops.to_nhwc(G["R"].to(d), dtype, 8)); torch.cuda.synchronize()
        errs = {}
        for k, ref in G["grad_checks"].items():
            v = dict(net.named_parameters())[k].grad.detach().float().cpu()
            if k.endswith("weight"): errs[k] = abs(float(v.norm() - ref[0])) / float(ref[0])
        print(name, dtype, "out", rel(nchw(out, 3), G["out"]), "dx", rel(nchw(x.grad, 3), G["dx"]), "worst dW norm err", max(errs.values()), max(errs, key=errs.get))
    # fp32 oracle with fp16-rounded input and weights: how much of the error is input/weight rounding?
    P = {k: v.half().float().requires_grad_(True) for k, v in O.synth_state_dict({k: torch.empty(G["shapes"][k]) for k in G["keys"]}, 0).items()}
    xx = G["x"].half().float().requires_grad_(True)
    o = O.resnet_generator(P, xx, c["n_blocks"]); (o * G["R"].half().float()).sum().backward()
    print(name, "oracle(fp16-rounded operands) vs golden: out", rel(o.detach(), G["out"]), "dx", rel(xx.grad, G["dx"]))

import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import joligen_amd
import torch
from joligen_amd import ops
d = torch.device("cuda", 0)
x = torch.randn(32, 64, 64, 256, device=d, dtype=torch.bfloat16, requires_grad=True)
g = torch.randn(32, 64, 64, 256, device=d, dtype=torch.bfloat16)
def f():
    y = x
    for _ in range(80):
        y = ops.group_norm(y, 256, None, None, None, 2, 1e-5)
    y.backward(g)
    return y
for _ in range(2):
    x.grad = None; f()
torch.cuda.synchronize()
xe = x.grad.clone()
gr = torch.cuda.CUDAGraph()
x.grad = None
with torch.cuda.graph(gr):
    ops.zero_pool_reset(d, True)
    yc = f()
ops.zero_pool_reset(d)
for i in range(3):
    x.grad.zero_()
    gr.replay()
    torch.cuda.synchronize()
    print(i, float(yc.float().abs().mean()), float(x.grad.float().abs().mean()), float(xe.float().abs().mean()), bool(torch.isfinite(x.grad).all()))

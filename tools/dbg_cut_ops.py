"""Dev tool (GPU box): op-level checks of the CUT glue ops against torch autograd."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from joligen_amd import ops
from joligen_amd.arena import ParamArena
from joligen_amd.modules.layers import JGConv2d, JGConvTranspose2d
d = torch.device("cuda:0"); dt = torch.float16
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
def nhwc(x): return x.permute(0, 2, 3, 1).contiguous()
def nchw(x): return x.permute(0, 3, 1, 2).contiguous()
g = torch.Generator().manual_seed(0)
# reflect pad
x = torch.randn(2, 16, 12, 10, generator=g).to(dt)
xr = x.float().requires_grad_(True); yr = F.pad(xr, (3, 3, 3, 3), mode="reflect"); R = torch.randn(yr.shape, generator=g).to(dt); yr.backward(R.float())
xd = nhwc(x).to(d).requires_grad_(True); y = ops.reflect_pad2d(xd, 3); y.backward(nhwc(R).to(d))
print("reflect fwd", rel(nchw(y).cpu(), yr.detach()), "bwd", rel(nchw(xd.grad).cpu(), xr.grad))
# act
for act, fn in ((ops.JG_ACT_TANH, torch.tanh), (ops.JG_ACT_LRELU, lambda t: F.leaky_relu(t, 0.2)), (ops.JG_ACT_RELU, F.relu)):
    xr = x.float().requires_grad_(True); yr = fn(xr); R = torch.randn(yr.shape, generator=g).to(dt); yr.backward(R.float())
    xd = nhwc(x).to(d).requires_grad_(True); y = ops.activation(xd, act); y.backward(nhwc(R).to(d))
    print("act", act, rel(nchw(y).cpu(), yr.detach()), rel(nchw(xd.grad).cpu(), xr.grad))
# instance norm + relu
xr = x.float().requires_grad_(True); yr = F.relu(F.instance_norm(xr)); R = torch.randn(yr.shape, generator=g).to(dt); yr.backward(R.float())
xd = nhwc(x).to(d).requires_grad_(True); y = ops.group_norm(xd, 16, None, None, None, ops.JG_ACT_RELU, 1e-5); y.backward(nhwc(R).to(d))
print("IN+relu", rel(nchw(y).cpu(), yr.detach()), rel(nchw(xd.grad).cpu(), xr.grad))
# convs
class M(nn.Module):
    def __init__(self, mod):
        super().__init__(); self.c = mod
for desc, mk, ref in (
    ("conv3 s2 p1", lambda: JGConv2d(16, 32, 3, padding=1, stride=2), lambda x, w, b: F.conv2d(x, w, b, stride=2, padding=1)),
    ("conv4 s2 p1", lambda: JGConv2d(16, 32, 4, padding=1, stride=2), lambda x, w, b: F.conv2d(x, w, b, stride=2, padding=1)),
    ("conv7 p0", lambda: JGConv2d(16, 8, 7, padding=0), lambda x, w, b: F.conv2d(x, w, b)),
    ("convT3 s2 p1 op1", lambda: JGConvTranspose2d(16, 32, 3, stride=2, padding=1, output_padding=1), lambda x, w, b: F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)),
):
    m = M(mk())
    with torch.no_grad():
        m.c.weight.copy_((torch.randn(m.c.weight.shape, generator=g) / math.sqrt(m.c.weight[0].numel())).to(dt).float())
        m.c.bias.copy_(torch.randn(m.c.bias.shape, generator=g) * 0.1)
    w0, b0 = m.c.weight.detach().clone(), m.c.bias.detach().clone()
    arena = ParamArena(m, d, dt, priority=()); arena.refresh()
    xx = torch.randn(2, 16, 12, 12, generator=g).to(dt)
    xr = xx.float().requires_grad_(True); wr = w0.clone().requires_grad_(True); br = b0.clone().requires_grad_(True)
    yr = ref(xr, wr, br); R = torch.randn(yr.shape, generator=g).to(dt); yr.backward(R.float())
    xd = nhwc(xx).to(d).requires_grad_(True); y = m.c(xd); y.backward(nhwc(R).to(d)); torch.cuda.synchronize()
    print(desc, "fwd", rel(nchw(y).cpu(), yr.detach()), "dx", rel(nchw(xd.grad).cpu(), xr.grad), "dw", rel(m.c.weight.grad.cpu(), wr.grad), "db", rel(m.c.bias.grad.cpu(), br.grad))

"""Dev tool (GPU box): ResnetBlock backward with a contiguous vs a non-contiguous upstream gradient (must agree)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch

import jg_oracle as O
from joligen_amd.modules.resnet_generator import ResnetGenerator

dtype = torch.float16
net = ResnetGenerator(3, 3, 16, n_blocks=2)
net.load_state_dict(O.synth_state_dict(net.state_dict(), 0))
net.jg_finalize(torch.device("cuda:0"), dtype)
net.arena.ensure_fresh()
blk = net.encoder.model[10]
g = torch.Generator(device="cuda").manual_seed(1)
h0 = torch.randn(1, 8, 8, 64, device="cuda", generator=g).to(dtype)
R = torch.randn(1, 8, 8, 64, device="cuda", generator=g).to(dtype)
res = []


def poison():
    """fill the caching allocator's free blocks with NaN so that any read of uninitialised memory shows up"""
    ts = [torch.full((n,), float("nan"), device="cuda", dtype=torch.float32) for n in (64, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 1 << 22, 1 << 24) for _ in range(8)]
    torch.cuda.synchronize()
    del ts


for mode in ("contig", "noncontig", "contig"):
    net.arena.zero_grad()
    poison()
    h = h0.clone().requires_grad_(True)
    out = blk(h)
    up = R.clone() if mode == "contig" else R.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
    assert up.is_contiguous() == (mode == "contig")
    poison()
    out.backward(up)
    torch.cuda.synchronize()
    res.append((h.grad.float().clone(), blk.conv_block[1].weight.grad.float().clone(), blk.conv_block[5].weight.grad.float().clone()))
    print(mode, [float(t.norm()) for t in res[-1]])
for i, n in enumerate(("dh", "dW1", "dW5")):
    print(n, "contig vs noncontig", float((res[0][i] - res[1][i]).norm() / res[1][i].norm()), "contig vs contig", float((res[0][i] - res[2][i]).norm() / res[2][i].norm()))

"""dev probe: which Python call sites issue device memcpy / memset inside one palette training step (torch.profiler, with stacks)"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench


class A:
    model, size, efficient, dtype, netG, netDs, force_exchange, batch = "palette", 256, 1, "bf16", "resnet", "basic", False, 8


model, opt = bench.build_model(A, 0, 0, 1)
data = bench.synth_batch(A.batch, 256, 7, torch.device("cuda:0"))
for _ in range(3):
    model.set_input(data)
    model.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.set_input(data)
    model.optimize_parameters()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if "emcpy" in n or "emset" in n or "copy_" == n or n in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::clone", "aten::contiguous", "aten::to"):
        st = [f for f in (e.stack or []) if "joligen_amd" in f or "bench.py" in f]
        cnt[(n, tuple(st[:3]))] += 1
for (n, st), c in cnt.most_common(40):
    print(c, n, " <- ".join(s.split("/")[-1] for s in st))

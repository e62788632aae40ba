"""Feasibility probe: capture the generator half of a CUT step (forward + losses + backward) in a hipGraph and replay it.
Dev tool (GPU box).  usage: python tools/cut_graph_probe.py"""
import argparse
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ns = argparse.Namespace(model="cut", netG="segformer_attn_conv", netDs="projected_d,basic", batch=16, size=256, dtype="bf16", efficient=1, force_exchange=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model, _ = bench.build_model(ns, 0, 0, 1)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(77)
batch = {"A": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev), "B": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)}
for p in (model.real_A_pool, model.real_B_pool, model.fake_B_pool):
    p.pool_size = 0
for _ in range(3):
    model.set_input(batch)
    model.optimize_parameters()
torch.cuda.synchronize()


def g_half():
    for network in model.model_names:
        model.set_requires_grad(getattr(model, "net" + network), network in model.group_G.networks_to_optimize)
    model.forward()
    model.compute_G_loss()
    model.loss_G_tot.backward()


def d_half():
    for network in model.model_names:
        model.set_requires_grad(getattr(model, "net" + network), network in model.group_D.networks_to_optimize)
    model.compute_D_loss()
    model.loss_D_tot.backward()


def timeit(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n


print("eager  G half: enqueue %.2f ms, wall %.2f ms" % timeit(g_half))
print("eager  D half: enqueue %.2f ms, wall %.2f ms" % timeit(d_half))
for name, fn in (("G", g_half), ("D", d_half)):
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        torch.cuda.synchronize()
        print("graph  %s half: enqueue %.2f ms, wall %.2f ms" % ((name,) + timeit(gr.replay)))
        print("   loss after replay:", float(model.loss_G_tot if name == "G" else model.loss_D_tot))
    except Exception as e:
        import traceback
        traceback.print_exc()
        print("capture of", name, "failed:", repr(e)[:300])
        torch.cuda.synchronize()

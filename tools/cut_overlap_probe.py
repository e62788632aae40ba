"""Sizing probe (GPU box): does the discriminator half of a CUT step (its own forward + backward, independent of the generator's
backward) shorten the step when it is replayed from a hipGraph on a SECOND stream while the generator's backward runs?  Timing only: the
captured graph keeps reading the capture-time fake / real tensors.  usage: python tools/cut_overlap_probe.py"""
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


def flags(group):
    for network in model.model_names:
        model.set_requires_grad(getattr(model, "net" + network), network in group.networks_to_optimize)


def g_fwd():
    flags(model.group_G)
    model.forward()
    model.compute_G_loss()


def g_bwd():
    model.loss_G_tot.backward()


def d_half():
    flags(model.group_D)
    model.compute_D_loss()
    model.loss_D_tot.backward()


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    d_half()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    d_half()
torch.cuda.synchronize()


def sequential_eager():
    g_fwd(); g_bwd(); d_half()


def sequential_graph():
    g_fwd(); g_bwd(); gr.replay()


def overlapped_graph():
    g_fwd()
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gr.replay()
    g_bwd()
    main.wait_stream(side)


def overlapped_eager():
    g_fwd()
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        d_half()
    flags(model.group_G)
    g_bwd()
    main.wait_stream(side)


for name, fn in (("sequential, eager D half", sequential_eager), ("sequential, D half from a graph", sequential_graph),
                 ("D half from a graph on a side stream under the G backward", overlapped_graph),
                 ("eager D half on a side stream under the G backward", overlapped_eager),
                 ("sequential, eager D half (again)", sequential_eager)):
    print("%-62s enqueue %.2f ms, wall %.2f ms (forward + backward of both halves, no optimizer steps)" % ((name,) + timeit(fn)))

"""Phase times of the CUT step (GPU box): each captured graph of the step replayed ALONE (synchronised around N replays), pairs of them
together, and the whole step.  Timing only -- replaying a graph out of order leaves gradients accumulated; nothing here is checked.
usage: python tools/cut_phase_probe.py [--netG segformer_attn_conv] [--netDs projected_d,basic] [--proj vitsmall]"""
import argparse
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--netG", default="segformer_attn_conv")
ap.add_argument("--netDs", default="projected_d,basic")
ap.add_argument("--proj", default="vitsmall")
ap.add_argument("--n", type=int, default=20)
a = ap.parse_args()
ns = argparse.Namespace(model="cut", netG=a.netG, netDs=a.netDs, batch=16, size=256, dtype="bf16", efficient=1, force_exchange=False, proj=a.proj)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model, _ = bench.build_model(ns, 0, 0, 1)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(77)
batch = {"A": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev), "B": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)}


def step():
    model.set_input(batch)
    model.optimize_parameters()


for _ in range(8):
    step()
torch.cuda.synchronize()
print("driver", model.step_driver, model.step_driver_note)
gst = next(iter(model._gg_graphs.values()))
dst = next(iter(model._dg_graphs.values()))
side = model._d_stream


def timeit(fn, n=a.n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def d_on_side():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dst["graph"].replay()
    torch.cuda.current_stream().wait_stream(side)


def fb():
    gst["fwd"].replay()
    gst["bwd"].replay()


def b_and_d():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dst["graph"].replay()
    gst["bwd"].replay()
    torch.cuda.current_stream().wait_stream(side)


def opt_only():
    model._group_finish(model.group_G)
    model._group_finish(model.group_D)


res = dict(
    graph_F=timeit(lambda: gst["fwd"].replay()),
    graph_B=timeit(lambda: gst["bwd"].replay()),
    graph_D=timeit(d_on_side),
    F_then_B=timeit(fb),
    B_with_D=timeit(b_and_d),
    optimizers=timeit(opt_only),
    step=timeit(step),
)
for k, v in res.items():
    print(f"{k:12s} {v:8.3f} ms")

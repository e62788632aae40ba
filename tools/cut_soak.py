"""Soak run of the benchmarked CUT selection (GPU box): N optimizer steps on fresh random batches with the default drivers (three captured graphs,
forked GAN branch), learning rates as configured; prints the losses every N / 10 steps and fails on a non-finite value or a dropped graph.
usage: python tools/cut_soak.py [steps]"""
import argparse
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ns = argparse.Namespace(model="cut", netG="segformer_attn_conv", netDs="projected_d,basic", batch=16, size=256, dtype="bf16", efficient=1, force_exchange=False, proj="vitsmall")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model, _ = bench.build_model(ns, 0, 0, 1)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(5)
t0 = time.perf_counter()
for it in range(steps):
    batch = {"A": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev), "B": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)}
    model.set_input(batch)
    model.optimize_parameters()
    if (it + 1) % max(1, steps // 10) == 0:
        vals = {k: float(getattr(model, "loss_" + k)) for k in ("G_tot", "G_NCE", "G_NCE_Y", "D_tot")}
        print(it + 1, model.step_driver, {k: round(v, 4) for k, v in vals.items()}, flush=True)
        assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
        assert model.step_driver == "graph+graphG", (model.step_driver, model.step_driver_note)
torch.cuda.synchronize()
pn = float(model._net("G_A").arena.p.float().norm())
assert pn == pn
print("ok: %d steps in %.1f s, |theta_G| = %.3f" % (steps, time.perf_counter() - t0, pn))

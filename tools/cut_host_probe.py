"""Is the CUT step host-bound?  Time the enqueue (no synchronisation inside) of N steps against their synchronised wall time; then the same
with the step replayed as one hipGraph (capture of set_input + optimize_parameters), if capture succeeds.  Dev tool (GPU box)."""
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
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
ns = argparse.Namespace(model="cut", netG=a.netG, netDs=a.netDs, batch=16, size=256, dtype="bf16", efficient=1, force_exchange=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model, _ = bench.build_model(ns, 0, 0, 1)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(77)
batch = {"A": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev), "B": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev)}


def step():
    model.set_input(batch)
    model.optimize_parameters()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / a.steps:.2f} ms/step (host), finished {1e3 * (t2 - t0) / a.steps:.2f} ms/step (wall), drain after last enqueue {1e3 * (t2 - t1):.2f} ms")
import cProfile
import pstats

pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
# which torch calls synchronise the host with the device inside a step?
import warnings as _w
torch.cuda.set_sync_debug_mode("warn")
with _w.catch_warnings(record=True) as rec:
    _w.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
seen = {}
for r in rec:
    if "synchroniz" in str(r.message):
        key = f"{r.filename}:{r.lineno}"
        seen[key] = seen.get(key, 0) + 1
print("synchronising torch calls in one step:", seen or "none")

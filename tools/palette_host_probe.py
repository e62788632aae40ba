"""how far ahead of the GPU the host runs in the palette step: host time of set_input + optimize_parameters (no synchronisation) against the
GPU-bound step time.  Dev tool (GPU box)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = argparse.Namespace(model="palette", efficient=1, size=256, batch=32, dtype="bf16", netG="resnet", netDs="basic", force_exchange=False, proj="efficientnet")
model, opt = bench.build_model(args, 0, 0, 1)
batch = bench.synth_batch(32, 256, 1234, torch.device("cuda:0"))
for _ in range(5):
    model.set_input(batch)
    model.optimize_parameters()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(30):
    h0 = time.perf_counter()
    model.set_input(batch)
    model.optimize_parameters()
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 30
host.sort()
print(f"wall per step {wall * 1e3:.2f} ms; host time inside set_input + optimize_parameters: median {host[15] * 1e3:.2f} ms, min {host[0] * 1e3:.2f}, max {host[-1] * 1e3:.2f} "
      f"(when the host runs ahead, its calls block on the launch queue and approach the wall time: the minimum is the host's own cost)")
# the host's own cost: one step enqueued onto an idle GPU after a synchronisation
own = []
for _ in range(5):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    model.set_input(batch)
    model.optimize_parameters()
    own.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
print("host enqueue time of one step onto an idle GPU (ms):", " ".join(f"{o * 1e3:.2f}" for o in own))

"""dev probe: eager vs hipGraph replay of forward+backward (B = 32), with the weight-gradient side stream on and off, same box"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from joligen_amd import parallel
from joligen_amd.modules import unet_exec


class A:
    model, size, efficient, dtype, netG, netDs, force_exchange, batch = "palette", 256, 1, "bf16", "resnet", "basic", False, 32


model, opt = bench.build_model(A, 0, 0, 1)
data = bench.synth_batch(32, 256, 7, torch.device("cuda:0"))
model.set_input(data)


def fb():
    model.compute_palette_loss()
    model.loss_G_tot.backward()
    for h in parallel._live(parallel.PRE_LAUNCH_HOOKS):      # join the side stream (what the optimizer launch does)
        h()


def timeit(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for side in (False, True):
    unet_exec.WGRAD_STREAM = side
    for _ in range(3):
        fb()
    torch.cuda.synchronize()
    te = timeit(fb)
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fb()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fb()
        torch.cuda.synchronize()
        tg = timeit(lambda: g.replay())
    except Exception as e:
        tg = float("nan")
        print("capture failed:", type(e).__name__, str(e)[:300])
    print(f"side stream {side}: eager fwd+bwd {te:.2f} ms, graph replay {tg:.2f} ms", flush=True)

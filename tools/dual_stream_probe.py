"""dev probe: two half-batch (B = 16) forward+backward passes replayed concurrently on two streams (complementary MFMA / HBM phases of the
two halves may overlap) against one B = 32 pass.  Two separate model instances: timing only."""
import os
import sys
import time

os.environ.setdefault("JG_WGRAD_STREAM", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


class A:
    model, size, efficient, dtype, netG, netDs, force_exchange = "palette", 256, 1, "bf16", "resnet", "basic", False
    batch = 32


def make(B, seed):
    A.batch = B
    model, opt = bench.build_model(A, 0, 0, 1)
    data = bench.synth_batch(B, 256, seed, torch.device("cuda:0"))
    model.set_input(data)

    def fb():
        model.compute_palette_loss()
        model.loss_G_tot.backward()

    for _ in range(2):
        fb()
    torch.cuda.synchronize()
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
    return model, g, s


def timeit(fn, n=8):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


m32, g32, s32 = make(32, 1)
print(f"B=32 one graph: {timeit(lambda: g32.replay()):.2f} ms", flush=True)
del m32, g32
torch.cuda.empty_cache()
ma, ga, sa = make(16, 2)
mb, gb, sb = make(16, 3)
print(f"B=16 one graph alone: {timeit(lambda: ga.replay()):.2f} ms", flush=True)


def both():
    with torch.cuda.stream(sa):
        ga.replay()
    with torch.cuda.stream(sb):
        gb.replay()


print(f"2 x B=16 on two streams: {timeit(both):.2f} ms", flush=True)

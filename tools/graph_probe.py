"""dev probe: does a micro-batched, graph-replayed forward+backward beat the eager batch-32 step?
usage: python tools/graph_probe.py [micro_batch] [n_micro]"""
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


def run(B, n_micro, graph):
    A.batch = B
    model, opt = bench.build_model(A, 0, 0, 1)
    data = bench.synth_batch(B, 256, 7, torch.device("cuda:0"))
    model.set_input(data)

    def fb():
        model.compute_palette_loss()
        (model.loss_G_tot / n_micro).backward()

    for _ in range(3):
        fb()
        model.optimizer_G.step(ema_beta=None)
    torch.cuda.synchronize()
    g = None
    if graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fb()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fb()
        torch.cuda.synchronize()
    times = []
    for it in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_micro):
            g.replay() if g is not None else fb()
        model.optimizer_G.step(ema_beta=None)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    ms = sorted(times)[len(times) // 2] * 1e3
    print(f"B={B} x{n_micro} graph={graph}: {ms:.2f} ms per {B * n_micro} images -> {B * n_micro / ms * 1e3:.1f} img/s", flush=True)
    del model
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run(32, 1, False)
    for mb in (16, 8, 4):
        try:
            run(mb, 32 // mb, True)
        except Exception as e:
            print("graph capture failed:", type(e).__name__, str(e)[:400], flush=True)
            break
    try:
        run(32, 1, True)
    except Exception as e:
        print("graph capture failed:", type(e).__name__, str(e)[:400], flush=True)

"""dev probe: GroupNorm backward (reduce -> coef -> apply) over the whole batch vs chunked by images (Infinity Cache residency)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from joligen_amd import _lib
from joligen_amd.ops import _st

L = _lib.lib()
dev = torch.device("cuda:0")
BF = _lib.JG_BF16


def run(B, HW, C, cb, reps=12, act=1, G=32):
    x = torch.randn(B, HW, C, device=dev).bfloat16()
    dy = torch.randn(B, HW, C, device=dev).bfloat16()
    dx = torch.empty_like(x)
    ab = torch.randn(B, C, 2, device=dev)
    mr = torch.rand(B, G, 2, device=dev) + 0.5
    red = torch.zeros(B, C, 2, device=dev)
    pqr = torch.empty(B, C, 3, device=dev)
    # flush: another big tensor pass between repetitions so that neither variant starts with x / dy on die
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    st = _st()

    def once():
        red.zero_()
        for b0 in range(0, B, cb):
            n = min(cb, B - b0)
            L.jg_gn_bwd_reduce(BF, x[b0].data_ptr(), dy[b0].data_ptr(), ab[b0].data_ptr(), red[b0].data_ptr(), n, HW, C, act, st)
            L.jg_gn_bwd_coef(red[b0].data_ptr(), None, None, None, 0, mr[b0].data_ptr(), pqr[b0].data_ptr(), None, None, None, 0, n, HW, C, G, st)
            L.jg_gn_bwd_apply(BF, x[b0].data_ptr(), dy[b0].data_ptr(), ab[b0].data_ptr(), pqr[b0].data_ptr(), dx[b0].data_ptr(), n, HW, C, act, st)

    once()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        junk.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        once()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    gb = 5 * x.numel() * 2 / 1e9
    us = tot / reps * 1e3
    print(f"B={B} HW={HW} C={C} chunk={cb:2d}: {us:8.1f} us  ({gb / (us * 1e-6) / 1e3:.2f} TB/s on 5N)", flush=True)
    return dx.float().sum().item()


for (B, HW, C) in ((32, 65536, 64), (32, 65536, 128), (32, 16384, 128), (32, 16384, 256), (32, 65536, 192)):
    ref = None
    for cb in (32, 16, 8, 4, 2):
        v = run(B, HW, C, cb)

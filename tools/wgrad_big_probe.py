"""wgrad_tn_big_kernel (256 x 256 tile) against the 128 x 128 tile on one 256 -> 256 point-wise weight gradient (131072 pixels), over split-K
factors.  usage: python tools/wgrad_big_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops

dev = torch.device("cuda", 0)
B, H, C = 32, 64, 256
x = torch.randn(B, H, H, C, device=dev, dtype=torch.bfloat16)
dy = torch.randn(B, H, H, C, device=dev, dtype=torch.bfloat16)
dw = torch.zeros(C, C, device=dev)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for big in (1, 0, 2, 11, 12, 13, 14):      # 1 DMA form, 0 the 128 x 128 tile, 2 register-staged form, 11 / 12 / 13 timing-only: no MFMAs / no fragment reads / no DMA, 14 the DMA instructions spread over the MFMA loop
    _lib.set_tuning("JG_WGRAD_BIG", big)
    for sk in ((16, 32, 64, 128, 256, 384, 512) if big in (0, 1) else (16, 128)):
        us = t(lambda: ops.wgrad_tn(dy, x, dw, B=B, H=H, W=H, Cin=C, Cout=C, R=1, S=1, pad=0, stride=1, Ho=H, Wo=H, lddy=C, ldx=C, lddw=C, splitk=sk))
        print("big %d splitk %3d: %6.1f us  %5.0f TFLOP/s  %.2f TB/s" % (big, sk, us, 2.0 * B * H * H * C * C / us / 1e6, 4.0 * B * H * H * C / us / 1e6))

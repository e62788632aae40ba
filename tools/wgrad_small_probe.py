"""Small weight-gradient launches (SegFormer / EfficientNet linear layers): time against the number of K slices and the output mode.
Dev tool (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops

d = torch.device("cuda:0")
dt = torch.bfloat16


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (B, H, W, Cin, Cout, k) in [(16, 1, 256, 160, 160, 1), (16, 1, 64, 256, 256, 1), (16, 1, 1024, 64, 64, 1), (16, 1, 4096, 32, 32, 1), (16, 8, 8, 256, 512, 4),
                                (16, 16, 16, 640, 160, 1), (16, 64, 64, 32, 128, 1), (16, 8, 8, 1024, 256, 1)]:
    pad = 1 if k == 4 else 0
    Ho, Wo = (H + 2 * pad - k) + 1, (W + 2 * pad - k) + 1
    x = torch.randn(B, H, W, Cin, device=d).to(dt)
    dy = torch.randn(B, Ho, Wo, Cout, device=d).to(dt)
    dw = torch.zeros(Cout, k, k, Cin, device=d)
    mpix = B * Ho * Wo
    tiles = ((Cout + 127) // 128) * ((k * k * Cin + 127) // 128)
    auto = ops._wgrad_splitk(tiles, mpix)
    row = []
    for sk in (1, 2, 4, 8, 16, 32, 64):
        if mpix // sk < 64:
            continue
        us = t(lambda: ops.wgrad_tn(dy, x, dw, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=k, S=k, pad=pad, stride=1, Ho=Ho, Wo=Wo, lddy=Cout, ldx=Cin,
                                    lddw=k * k * Cin, splitk=sk))
        row.append(f"sk{sk}:{us:6.1f}")
    us_store = t(lambda: ops.wgrad_tn(dy, x, dw, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=k, S=k, pad=pad, stride=1, Ho=Ho, Wo=Wo, lddy=Cout, ldx=Cin,
                                      lddw=k * k * Cin, splitk=1, out_mode=_lib.JG_OUT_STORE_F32))
    print(f"{(B, H, W, Cin, Cout, k)} mpix {mpix} tiles {tiles} auto sk {auto} | " + " ".join(row) + f" | store sk1:{us_store:6.1f} us")

"""timing of the 7x7 content head of the CUT generators (64 -> 32 at 256^2 x 32 images, forward; 32 -> 64 with pad 6, input gradient) on the
halo-resident kernel (JG_CONV_KXK=1) and on the im2col kernel.  Dev tool (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from joligen_amd import _lib, ops
from tools.conv_bench import timeit

d = torch.device("cuda:0")
dt = torch.bfloat16
for name, (B, Hin, Cin, Cout, pad) in {"forward 64->32 (262^2 -> 256^2)": (32, 262, 64, 32, 0), "input gradient 32->64 (256^2 -> 262^2, pad 6)": (32, 256, 32, 64, 6)}.items():
    Ho = Hin + 2 * pad - 6
    x = torch.randn(B, Hin, Hin, Cin, device=d).to(dt)
    w = (torch.randn(Cout, 7, 7, Cin, device=d) / (49 * Cin) ** 0.5).to(dt)
    y = torch.empty(B, Ho, Ho, Cout, device=d, dtype=dt)
    geo = dict(B=B, H=Hin, W=Hin, Cin=Cin, Cout=Cout, R=7, S=7, pad=pad, stride=1, Ho=Ho, Wo=Ho, ldx=Cin, ldw=49 * Cin, ldy=Cout)
    flops = 2.0 * B * Ho * Ho * Cout * 49 * Cin
    line = name + ":"
    for v in (1, 2, 0):
        _lib.set_tuning("JG_CONV_KXK", v)
        ops.conv_nt(x, w, y, **geo)
        t = timeit(lambda: ops.conv_nt(x, w, y, **geo), reps=10)
        line += f"  {_lib.lib().jg_last_kernel().decode()} {t * 1e6:7.1f} us {flops / t / 1e12:6.0f} TFLOP/s"
    _lib.set_tuning("JG_CONV_KXK", 1)
    print(line, flush=True)

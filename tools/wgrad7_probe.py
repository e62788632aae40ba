"""7x7 weight gradient of the CUT content head (x [32,262,262,64] -> dy [32,256,256,32]): wgrad_kxk.hip against the im2col kernel.  Dev tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops

d = torch.device("cuda:0")
B, H, W, Cin, Cout, pad = 32, 262, 262, 64, 32, 0
Ho, Wo = H - 6, W - 6
x = torch.randn(B, H, W, Cin, device=d).to(torch.bfloat16)
dy = torch.randn(B, Ho, Wo, Cout, device=d).to(torch.bfloat16)
dw = torch.zeros(Cout, 7, 7, Cin, device=d)
sk = ops._wgrad_splitk(((Cout + 127) // 128) * ((49 * Cin + 127) // 128), B * Ho * Wo)
for var in (4, 3, 4, 3):
    _lib.set_tuning("JG_WGRAD_VARIANT", var)
    f = lambda: ops.wgrad_tn(dy, x, dw, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=7, S=7, pad=pad, stride=1, Ho=Ho, Wo=Wo, lddy=Cout, ldx=Cin, lddw=49 * Cin, splitk=sk)
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"variant {var} {_lib.lib().jg_last_kernel().decode() or 'wgrad_tn_tr'}: {us:8.1f} us  {2.0 * B * Ho * Wo * Cout * 49 * Cin / us / 1e6:7.1f} TFLOP/s")

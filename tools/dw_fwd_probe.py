"""depth-wise 3x3 forward: run form (JG_DW_RUN=1) against the per-pixel form, mobile ResNet block and MiT MixFFN shapes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from joligen_amd import _lib
from joligen_amd.ops import _st
L = _lib.lib()
d = torch.device("cuda:0")
BF = _lib.JG_BF16
for (B, H, C, refl, gelu) in ((32, 64, 256, 1, 0), (16, 64, 256, 1, 0), (64, 64, 128, 0, 1), (64, 32, 256, 0, 1), (64, 16, 640, 0, 1), (64, 8, 1024, 0, 1)):
    x = torch.randn(B, H, H, C, device=d).bfloat16(); y = torch.empty_like(x); pre = torch.empty_like(x)
    w = torch.randn(C, 9, device=d); b = torch.randn(C, device=d)
    outs = []
    for run in (1, 0):
        _lib.set_tuning("JG_DW_RUN", run)
        f = lambda: L.jg_dwconv3x3_fwd_pad(BF, x.data_ptr(), w.data_ptr(), b.data_ptr(), pre.data_ptr() if gelu else None, y.data_ptr(), B, H, H, C, gelu, refl, _st())
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = x.numel() * 2 * (2 + (1 if gelu else 0)) / 1e6
        outs.append(y.clone())
        print(f"B={B} H={H} C={C} reflect={refl} run={run}: {us:7.1f} us  ({mb:.0f} MB -> {mb / us * 1e-3:.2f} TB/s)", flush=True)
    print("   max |run - pixel| =", float((outs[0].float() - outs[1].float()).abs().max()))

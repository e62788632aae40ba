"""dev probe: depth-wise 3x3 backward (du, dw, dbias, dx) per MiT stage shape"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from joligen_amd import _lib
from joligen_amd.ops import _st
L = _lib.lib()
d = torch.device("cuda:0")
BF = _lib.JG_BF16
for (B, H, C) in ((32, 64, 128), (32, 32, 256), (32, 16, 640), (32, 8, 1024), (16, 64, 128)):
    x = torch.randn(B, H, H, C, device=d).bfloat16(); pre = torch.randn_like(x); dy = torch.randn_like(x)
    du = torch.empty_like(x); dx = torch.empty_like(x)
    w = torch.randn(C, 9, device=d); dw = torch.zeros(C, 9, device=d); db = torch.zeros(C, device=d)
    def run(with_dx):
        L.jg_dwconv3x3_bwd(BF, x.data_ptr(), pre.data_ptr(), dy.data_ptr(), w.data_ptr(), du.data_ptr(), dx.data_ptr() if with_dx else None,
                           dw.data_ptr(), db.data_ptr(), B, H, H, C, 1, _st())
    for with_dx in (False, True):
        for _ in range(3): run(with_dx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(with_dx)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = x.numel() * 2 * (4 + (2 if with_dx else 0)) / 1e6
        print(f"B={B} H={H} C={C} dx={with_dx}: {us:7.1f} us  ({mb:.0f} MB algorithmic -> {mb / us * 1e-3:.2f} TB/s)", flush=True)

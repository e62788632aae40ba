"""the ViT projector's GEMMs (16 x 257 tokens) on the tile variants of the generic LDS-DMA kernel.  Dev tool (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from joligen_amd import _lib, ops
from tools.conv_bench import timeit

d, dt = torch.device("cuda:0"), torch.bfloat16
M = 16 * 257
ws = torch.empty(32 << 20, device=d, dtype=torch.uint8)
for K, N in ((384, 1536), (1536, 384), (384, 1152), (384, 384), (1152, 384)):
    x = torch.randn(M, K, device=d).to(dt)
    w = (torch.randn(N, K, device=d) / K ** 0.5).to(dt)
    y = torch.empty(M, N, device=d, dtype=dt)
    bias = torch.randn(N, device=d)
    geo = dict(B=16, H=1, W=257, Cin=K, Cout=N, R=1, S=1, pad=0, stride=1, Ho=1, Wo=257, ldx=K, ldw=K, ldy=N)
    line = f"{K:5d} -> {N:5d}:"
    for var, small in ((6, 1), (6, 0), (6, 2), (4, 0), (5, 0), (2, 0)):
        _lib.set_tuning("JG_CONV_VARIANT", var)
        _lib.set_tuning("JG_CONV_SMALL_TILE", small)
        ops.conv_nt(x, w, y, bias=bias, **geo)
        t = timeit(lambda: ops.conv_nt(x, w, y, bias=bias, **geo), reps=20)
        line += f"  v{var}/s{small} {(_lib.lib().jg_last_kernel().decode() or 'default')[-22:]:>22s} {t * 1e6:6.1f} us {2.0 * M * K * N / t / 1e12:5.0f} TF"
    _lib.set_tuning("JG_CONV_VARIANT", 6)
    _lib.set_tuning("JG_CONV_SMALL_TILE", 1)
    print(line, flush=True)

"""timing-only ablation (JG_HALO_DBG bits: 1 no atomic epilogue, 4 per-tile LDS-DMA for the first tile only, 8 no per-tile barriers, 2 no MFMAs /
fragment reads) of wgrad3x3_halo_kernel configurations.  Results are WRONG under these bits: timing only.  Dev tool (GPU box).
usage: python tools/wgrad_cfg_ablate.py [cfgs=1,4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops
from joligen_amd.ops import JG_OUT_ATOMIC_F32
from tools.conv_bench import timeit

CFGS = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "4"])]
B, dt, d = 32, torch.bfloat16, torch.device("cuda:0")
for Cin, Cout, S in ((512, 512, 32), (64, 64, 256), (256, 256, 64)):
    x = (torch.rand(B, S, S, Cin, device=d) * 2 - 1).to(dt)
    dy = (torch.rand(B, S, S, Cout, device=d) * 2 - 1).to(dt)
    dw = torch.zeros(Cout, 3, 3, Cin, device=d, dtype=torch.float32)
    geo = dict(B=B, H=S, W=S, R=3, S=3, pad=1, stride=1, Ho=S, Wo=S)

    def run():
        ops.wgrad_tn(dy, x, dw, Cin=Cin, Cout=Cout, lddy=Cout, ldx=Cin, lddw=9 * Cin, splitk=1, out_mode=JG_OUT_ATOMIC_F32, **geo)
    mf = 2.0 * B * S * S * Cout * 9 * Cin / (16 * 16 * 32 * 2) * 16 / 1024 / 2.1e9 * 1e6
    print(f"{Cin}->{Cout} @ {S}: MFMA pipe time at 2.1 GHz = {mf:.1f} us")
    for c in CFGS:
        _lib.set_tuning("JG_WGRAD_HALO_CFG", c)
        line = f"  cfg{c}:"
        for dbg, tag in ((0, "full"), (1, "-epi"), (5, "-epi-dma"), (13, "-epi-dma-bar"), (2, "no mfma"), (6, "no mfma no dma"), (7, "prologue+walk only")):
            _lib.set_tuning("JG_HALO_DBG", dbg)
            run()
            line += f"  {tag} {timeit(run, reps=8) * 1e6:6.1f}"
        _lib.set_tuning("JG_HALO_DBG", 0)
        print(line, flush=True)
    _lib.set_tuning("JG_WGRAD_HALO_CFG", 0)

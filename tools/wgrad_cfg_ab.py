"""A/B of the tile configurations of wgrad3x3_halo_kernel (JG_WGRAD_HALO_CFG through jg_set_tuning) on the 3x3 layers of BASELINE configs[1]
that the 16-row / 64-co tile serves: agreement with the first configuration and TFLOP/s.  Dev tool (GPU box).

usage: python tools/wgrad_cfg_ab.py [cfgs, default 1,4] [--all]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops
from joligen_amd.ops import JG_OUT_ATOMIC_F32
from tools.conv_bench import SHAPES, timeit

args = [a for a in sys.argv[1:] if not a.startswith("--")]
CFGS = [int(c) for c in (args[0].split(",") if args else ["1", "4"])]
B, dt, d = 32, torch.bfloat16, torch.device("cuda:0")
shapes = [s for s in SHAPES if s[2] == 3 and s[0] % 64 == 0 and s[1] % 64 == 0]
tot = {c: 0.0 for c in CFGS}
auto_tot = 0.0
print("  Cin  Cout    S cnt | auto kernel                                   ms | " + " | ".join(f"cfg{c}   ms    TF   err_w" for c in CFGS))
for Cin, Cout, k, S, cnt in shapes:
    g = torch.Generator(device=d).manual_seed(Cin * 7 + Cout)
    x = (torch.rand(B, S, S, Cin, device=d, generator=g) * 2 - 1).to(dt)
    dy = (torch.rand(B, S, S, Cout, device=d, generator=g) * 2 - 1).to(dt)
    geo = dict(B=B, H=S, W=S, R=k, S=k, pad=1, stride=1, Ho=S, Wo=S)
    flops = 2.0 * B * S * S * Cout * 9 * Cin
    dw = torch.zeros(Cout, 3, 3, Cin, device=d, dtype=torch.float32)
    db = torch.zeros(Cout, device=d, dtype=torch.float32)

    def run():
        ops.wgrad_tn(dy, x, dw, Cin=Cin, Cout=Cout, lddy=Cout, ldx=Cin, lddw=9 * Cin, splitk=1, out_mode=JG_OUT_ATOMIC_F32, dbias=db, **geo)
    _lib.set_tuning("JG_WGRAD_HALO_CFG", 0)
    run()
    name = _lib.lib().jg_last_kernel().decode()
    ta = timeit(run)
    auto_tot += ta * cnt
    if "16 rows" not in name and "--all" not in sys.argv:
        continue
    line = f"{Cin:5d} {Cout:5d} {S:4d} {cnt:3d} | {name:42s} {ta * 1e3:6.3f} |"
    ref = None
    for c in CFGS:
        _lib.set_tuning("JG_WGRAD_HALO_CFG", c)
        dw.zero_()
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref, ew = dw.clone(), 0.0
        else:
            ew = float((dw - ref).norm() / ref.norm())
        t = timeit(run)
        tot[c] += t * cnt
        line += f" {t * 1e3:9.3f} {flops / t / 1e12:5.0f} {ew:7.1e} |"
    _lib.set_tuning("JG_WGRAD_HALO_CFG", 0)
    print(line, flush=True)
for c in CFGS:
    print(f"TOTAL cfg{c}: {tot[c] * 1e3:.3f} ms per step-equivalent over the listed layers")
print(f"TOTAL auto (all 3x3 layers): {auto_tot * 1e3:.3f} ms")

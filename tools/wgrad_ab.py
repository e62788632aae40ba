"""A/B of the weight-gradient kernel variants (JG_WGRAD_VARIANT) on the 3x3 layers of BASELINE
configs[1]: agreement with variant 2 (the parity-tested im2col kernel) and TFLOP/s.  Dev tool (GPU box).

usage: python tools/wgrad_ab.py [variants, default 2,4] [--quick]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import ops
from joligen_amd.ops import JG_OUT_ATOMIC_F32
from tools.conv_bench import SHAPES, timeit

args = [a for a in sys.argv[1:] if not a.startswith("--")]
VARIANTS = args[0].split(",") if args else ["2", "4"]
QUICK = "--quick" in sys.argv
B = 32
dt = torch.bfloat16
d = torch.device("cuda:0")
tot = {v: 0.0 for v in VARIANTS}
flops_tot = 0.0
shapes = [s for s in SHAPES if s[2] == 3 and s[0] % 64 == 0 and s[1] % 64 == 0]
if QUICK:
    shapes = shapes[:5]
print("  Cin  Cout k    S cnt | " + " | ".join(f"v{v:>4} ms    TF  err_w   err_b" for v in VARIANTS))
for Cin, Cout, k, S, cnt in shapes:
    g = torch.Generator(device=d).manual_seed(Cin * 7 + Cout)
    x = (torch.rand(B, S, S, Cin, device=d, generator=g) * 2 - 1).to(dt)
    dy = (torch.rand(B, S, S, Cout, device=d, generator=g) * 2 - 1).to(dt)
    geo = dict(B=B, H=S, W=S, R=k, S=k, pad=1, stride=1, Ho=S, Wo=S)
    flops = 2.0 * B * S * S * Cout * k * k * Cin
    ktot = k * k * Cin
    tiles = ((Cout + 127) // 128) * ((ktot + 127) // 128)
    splitk = ops._wgrad_splitk(tiles, B * S * S)
    line = f"{Cin:5d} {Cout:5d} {k} {S:4d} {cnt:3d} |"
    ref = refb = None
    for v in VARIANTS:
        vv = v.split(":")
        os.environ["JG_WGRAD_VARIANT"] = vv[0]
        os.environ["JG_WGRAD_HALO_CFG"] = vv[1] if len(vv) > 1 else "0"
        dw = torch.zeros(Cout, k, k, Cin, device=d, dtype=torch.float32)
        db = torch.zeros(Cout, device=d, dtype=torch.float32)

        def run():
            ops.wgrad_tn(dy, x, dw, Cin=Cin, Cout=Cout, lddy=Cout, ldx=Cin, lddw=ktot, splitk=splitk,
                         out_mode=JG_OUT_ATOMIC_F32, dbias=db, **geo)
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref, refb = dw.clone(), db.clone()
            ew = eb = 0.0
        else:
            ew = float((dw - ref).norm() / ref.norm())
            eb = float((db - refb).norm() / refb.norm())
        t = timeit(run)
        tot[v] += t * cnt
        line += f" {t * 1e3:6.3f} {flops / t / 1e12:5.0f} {ew:7.1e} {eb:7.1e} |"
    flops_tot += flops * cnt
    print(line, flush=True)
for v in VARIANTS:
    print(f"TOTAL v{v}: {tot[v] * 1e3:.2f} ms per step-equivalent, {flops_tot / tot[v] / 1e12:.0f} TFLOP/s")

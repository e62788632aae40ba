"""A/B of conv_nt kernel variants (JG_CONV_VARIANT) on the 3x3 layers of BASELINE configs[1]:
numerical agreement of every variant with variant 3 (the parity-tested LDS-DMA im2col kernel)
and TFLOP/s from HIP events.  Dev tool (GPU box).

usage: python tools/conv_ab.py [variants, default 3,6] [--quick]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import ops
from tools.conv_bench import SHAPES, timeit

args = [a for a in sys.argv[1:] if not a.startswith("--")]
VARIANTS = args[0].split(",") if args else ["3", "6"]
QUICK = "--quick" in sys.argv
ONLY = [tuple(int(v) for v in a[7:].split("x")) for a in sys.argv if a.startswith("--only=")]  # --only=CinxCoutxS
B = 32
dt = torch.bfloat16
d = torch.device("cuda:0")
tot = {v: 0.0 for v in VARIANTS}
flops_tot = 0.0
shapes = [s for s in SHAPES if s[2] == 3 and s[0] % 64 == 0 and s[1] % 64 == 0]
# dgrad shapes = forward shapes with the channel counts swapped
seen = set()
allshapes = []
for Cin, Cout, k, S, cnt in shapes:
    for ci, co in ((Cin, Cout), (Cout, Cin)):
        key = (ci, co, S)
        if key in seen:
            for s in allshapes:
                if (s[0], s[1], s[3]) == key:
                    s[4] += cnt
            continue
        seen.add(key)
        allshapes.append([ci, co, k, S, cnt])
if QUICK:
    allshapes = allshapes[:6]
if ONLY:
    allshapes = [s for s in allshapes if (s[0], s[1], s[3]) in ONLY]
print("  Cin  Cout k    S cnt | " + " | ".join(f"v{v:>2} ms    TF  relerr" for v in VARIANTS))
for Cin, Cout, k, S, cnt in allshapes:
    g = torch.Generator(device=d).manual_seed(Cin * 7 + Cout)
    x = (torch.rand(B, S, S, Cin, device=d, generator=g) * 2 - 1).to(dt)
    w = ((torch.rand(Cout, k, k, Cin, device=d, generator=g) * 2 - 1) / (k * Cin ** 0.5)).to(dt)
    bias = torch.rand(Cout, device=d, generator=g)
    res = (torch.rand(B, S, S, Cout, device=d, generator=g) * 2 - 1).to(dt)
    geo = dict(B=B, H=S, W=S, R=k, S=k, pad=1, stride=1, Ho=S, Wo=S)
    flops = 2.0 * B * S * S * Cout * k * k * Cin
    line = f"{Cin:5d} {Cout:5d} {k} {S:4d} {cnt:3d} |"
    ref = None
    for v in VARIANTS:
        vv = v.split(":")
        os.environ["JG_CONV_VARIANT"] = vv[0]
        os.environ["JG_HALO_CFG"] = vv[1] if len(vv) > 1 else "0"
        y = torch.zeros(B, S, S, Cout, device=d, dtype=dt)
        ops.conv_nt(x, w, y, Cin=Cin, Cout=Cout, ldx=Cin, ldw=k * k * Cin, ldy=Cout, bias=bias, res=res, ldres=Cout,
                    res_scale=0.7071, **geo)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.float()
            err = 0.0
        else:
            err = float((y.float() - ref).norm() / ref.norm())
        t = timeit(lambda: ops.conv_nt(x, w, y, Cin=Cin, Cout=Cout, ldx=Cin, ldw=k * k * Cin, ldy=Cout, **geo))
        tot[v] += t * cnt
        line += f" {t * 1e3:6.3f} {flops / t / 1e12:5.0f} {err:7.1e} |"
    flops_tot += flops * cnt
    print(line, flush=True)
for v in VARIANTS:
    print(f"TOTAL v{v}: {tot[v] * 1e3:.2f} ms per step-equivalent, {flops_tot / tot[v] / 1e12:.0f} TFLOP/s")

"""A/B microbenchmark of conv_nt kernel variants (JG_CONV_VARIANT) at the BASELINE configs[1]
forward shapes (dgrad uses the same kernel on transposed channel counts).  Dev tool (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import ops
from tools.conv_bench import SHAPES, timeit

VARIANTS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2", "3", "4", "5"]
B = 32
dt = torch.bfloat16
d = torch.device("cuda:0")
tot = {v: 0.0 for v in VARIANTS}
best = 0.0
flops_tot = 0.0
print("  Cin  Cout k    S cnt | " + " | ".join(f"v{v:>2} ms    TF" for v in VARIANTS))
for Cin, Cout, k, S, cnt in SHAPES:
    pad = k // 2
    x = torch.randn(B, S, S, Cin, device=d).to(dt)
    w = (torch.randn(Cout, k, k, Cin, device=d) / (k * Cin ** 0.5)).to(dt)
    y = torch.empty(B, S, S, Cout, device=d, dtype=dt)
    geo = dict(B=B, H=S, W=S, R=k, S=k, pad=pad, stride=1, Ho=S, Wo=S)
    flops = 2.0 * B * S * S * Cout * k * k * Cin
    line = f"{Cin:5d} {Cout:5d} {k} {S:4d} {cnt:3d} |"
    ts = []
    for v in VARIANTS:
        os.environ["JG_CONV_VARIANT"] = v
        t = timeit(lambda: ops.conv_nt(x, w, y, Cin=Cin, Cout=Cout, ldx=Cin, ldw=k * k * Cin, ldy=Cout, **geo))
        ts.append(t)
        tot[v] += t * cnt
        line += f" {t * 1e3:6.3f} {flops / t / 1e12:5.0f} |"
    best += min(ts) * cnt
    flops_tot += flops * cnt
    print(line, flush=True)
for v in VARIANTS:
    print(f"TOTAL v{v}: {tot[v] * 1e3:.2f} ms, {flops_tot / tot[v] / 1e12:.0f} TFLOP/s")
print(f"TOTAL best-per-layer: {best * 1e3:.2f} ms, {flops_tot / best / 1e12:.0f} TFLOP/s")

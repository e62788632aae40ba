"""A/B of the software-pipelined K loop of conv3x3_halo_kernel (JG_HALO_PIPE 1, csrc/mfma_pipe.h) against the compiler-scheduled loop
(JG_HALO_PIPE 0) on the 3x3 shapes of BASELINE configs[1] (batch 32, 256x256): bit-identity of the outputs, interleaved timing rounds in
ONE process (HIP events), TFLOP/s.  Dev tool (GPU box).

usage: python tools/halo_pipe_ab.py [--rounds 5] [--switch JG_HALO_PIPE] [--values 0,1] [--min-ch 128]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops
from tools.conv_bench import SHAPES

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--switch", default="JG_HALO_PIPE")
ap.add_argument("--values", default="0,1")
ap.add_argument("--min-ch", type=int, default=64)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--res", type=int, default=0)
ap.add_argument("--dbg", type=int, default=0, help="JG_HALO_DBG ablation bits (timing only, results are wrong)")
ap.add_argument("--only", default="", help="comma list of Cin-Cout-S")
args = ap.parse_args()
VALUES = [int(v) for v in args.values.split(",")]
B = 32
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
d = torch.device("cuda:0")

shapes = {}
for Cin, Cout, k, S, cnt in SHAPES:
    if k != 3 or Cin % 64 or Cout % 64:
        continue
    for ci, co in ((Cin, Cout), (Cout, Cin)):      # forward and input gradient (channel counts swapped)
        if min(ci, co) < args.min_ch:
            continue
        shapes[(ci, co, S)] = shapes.get((ci, co, S), 0) + cnt


def time_once(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


if args.dbg:
    _lib.set_tuning("JG_HALO_DBG", args.dbg)
tot = {v: 0.0 for v in VALUES}
ftot = 0.0
print(f"switch {args.switch}; columns per value: best-of-rounds us, TFLOP/s | identical")
for (Cin, Cout, S), cnt in sorted(shapes.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2] ** 2):
    if args.only and f"{Cin}-{Cout}-{S}" not in args.only.split(","):
        continue
    g = torch.Generator(device=d).manual_seed(Cin * 7 + Cout)
    x = (torch.rand(B, S, S, Cin, device=d, generator=g) * 2 - 1).to(dt)
    w = ((torch.rand(Cout, 3, 3, Cin, device=d, generator=g) * 2 - 1) / (3 * Cin ** 0.5)).to(dt)
    bias = torch.rand(Cout, device=d, generator=g)
    res = (torch.rand(B, S, S, Cout, device=d, generator=g) * 2 - 1).to(dt) if args.res else None
    st = torch.zeros(B, 16, Cout, 2, device=d)
    geo = dict(B=B, H=S, W=S, R=3, S=3, pad=1, stride=1, Ho=S, Wo=S)
    flops = 2.0 * B * S * S * Cout * 9 * Cin
    outs = {}
    ys = {v: torch.zeros(B, S, S, Cout, device=d, dtype=dt) for v in VALUES}

    def run(v):
        ops.conv_nt(x, w, ys[v], Cin=Cin, Cout=Cout, ldx=Cin, ldw=9 * Cin, ldy=Cout, bias=bias, res=res, ldres=Cout if res is not None else 0,
                    res_scale=0.7071, stats=st, ldstats=Cout, stats_slots=16, **geo)

    best = {v: 1e9 for v in VALUES}
    for v in VALUES:
        _lib.set_tuning(args.switch, v)
        run(v)
    torch.cuda.synchronize()
    same = all(torch.equal(ys[v], ys[VALUES[0]]) for v in VALUES)
    err = max(float((ys[v].float() - ys[VALUES[0]].float()).abs().max()) for v in VALUES)
    for _ in range(args.rounds):
        for v in VALUES:
            _lib.set_tuning(args.switch, v)
            run(v)
            best[v] = min(best[v], time_once(lambda: run(v), args.reps))
    line = f"{Cin:5d}->{Cout:5d} @{S:3d} x{cnt:2d} |"
    for v in VALUES:
        line += f" v{v}: {best[v] * 1e6:7.1f} us {flops / best[v] / 1e12:6.0f} TF |"
        tot[v] += best[v] * cnt
    ftot += flops * cnt
    print(line + (" identical" if same else f" DIFFERENT max|d|={err:.3e}"), flush=True)
for v in VALUES:
    print(f"TOTAL v{v}: {tot[v] * 1e3:.3f} ms per step-equivalent, {ftot / tot[v] / 1e12:.0f} TFLOP/s")
_lib.set_tuning(args.switch, VALUES[-1])

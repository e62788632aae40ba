"""A/B of a dispatch switch of the halo-resident weight-gradient kernel (default JG_WGRAD_PIPE 0 / 1) on the 3x3 shapes of BASELINE
configs[1] (batch 32, 256x256): agreement of the fp32 gradients (atomics: summation order differs run to run, so norm-wise), interleaved
timing rounds in one process, TFLOP/s.  Dev tool (GPU box).   usage: python tools/wgrad_pipe_ab.py [--switch NAME] [--values 0,1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops
from tools.conv_bench import SHAPES

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--switch", default="JG_WGRAD_PIPE")
ap.add_argument("--values", default="0,1")
ap.add_argument("--only", default="")
ap.add_argument("--dbg", type=int, default=0)
args = ap.parse_args()
VALUES = [int(v) for v in args.values.split(",")]
B, dt, d = 32, torch.bfloat16, torch.device("cuda:0")


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
for Cin, Cout, k, S, cnt in SHAPES:
    if k != 3 or Cin % 64 or Cout % 64:
        continue
    if args.only and f"{Cin}-{Cout}-{S}" not in args.only.split(","):
        continue
    g = torch.Generator(device=d).manual_seed(Cin * 7 + Cout)
    x = (torch.rand(B, S, S, Cin, device=d, generator=g) * 2 - 1).to(dt)
    dy = (torch.rand(B, S, S, Cout, device=d, generator=g) * 2 - 1).to(dt)
    geo = dict(B=B, H=S, W=S, R=3, S=3, pad=1, stride=1, Ho=S, Wo=S)
    flops = 2.0 * B * S * S * Cout * 9 * Cin
    tiles = ((Cout + 127) // 128) * ((9 * Cin + 127) // 128)
    splitk = ops._wgrad_splitk(tiles, B * S * S)
    dws = {v: torch.zeros(Cout, 3, 3, Cin, device=d) for v in VALUES}
    dbs = {v: torch.zeros(Cout, device=d) for v in VALUES}

    def run(v):
        ops.wgrad_tn(dy, x, dws[v], Cin=Cin, Cout=Cout, lddy=Cout, ldx=Cin, lddw=9 * Cin, dbias=dbs[v], Cin_out=Cin, Cout_out=Cout, splitk=splitk,
                     dbias_scale=1.0, **geo)

    for v in VALUES:
        _lib.set_tuning(args.switch, v)
        run(v)
    torch.cuda.synchronize()
    ref, rb = dws[VALUES[0]], dbs[VALUES[0]]
    err = max(float((dws[v] - ref).norm() / ref.norm()) for v in VALUES)
    errb = max(float((dbs[v] - rb).norm() / rb.norm()) for v in VALUES)
    best = {v: 1e9 for v in VALUES}
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
    print(line + f" rel dw {err:.1e} db {errb:.1e}", flush=True)
for v in VALUES:
    print(f"TOTAL v{v}: {tot[v] * 1e3:.3f} ms per step-equivalent, {ftot / tot[v] / 1e12:.0f} TFLOP/s")
_lib.set_tuning(args.switch, VALUES[-1])

"""Ablation timing of the halo-resident 3x3 convolution at the bench's layer shapes (dev tool, GPU box).

For each shape: the full kernel, then with the epilogue / the MFMA + LDS fragment reads / the halo DMA switched off
(JG_HALO_DBG bits 1 / 2 / 4 through jg_set_tuning -- wrong results, timing only).  Epilogue options as in the UNet
schedule: bias + fused GroupNorm statistics (+ residual with --res).  Operands rotate through --rotate sets so that the
256 MB MALL does not hold them between launches."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops

SHAPES = [  # Cin, Cout, S, launches per step (forward + input gradient)
    (64, 64, 256, 14), (128, 128, 128, 14), (256, 256, 64, 14), (512, 512, 32, 22), (128, 128, 256, 2), (256, 256, 128, 2),
    (128, 64, 256, 2), (64, 128, 256, 2), (512, 512, 64, 2), (1024, 512, 32, 2), (64, 192, 256, 1), (192, 64, 256, 1),
]


def timeit(fn, reps=6, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rotate", type=int, default=3)
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--dbg", default="0,1,2,4,6")
    ap.add_argument("--cfgs", default="0", help="JG_HALO_CFG values to compare")
    ap.add_argument("--persist", default="", help="JG_PERSIST64 values to compare (empty: leave the default)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dt, d, B = torch.bfloat16, torch.device("cuda:0"), args.batch
    dbgs = [int(v) for v in args.dbg.split(",")]
    cfgs = [int(v) for v in args.cfgs.split(",")]
    persists = [int(v) for v in args.persist.split(",")] if args.persist else [None]
    print(f"{'Cin':>5} {'Cout':>5} {'S':>4} cnt cfg per | " + " | ".join(f"dbg{v} us    TF" for v in dbgs))
    tot = {}
    for Cin, Cout, S, cnt in SHAPES:
        if args.only and f"{Cin}-{Cout}-{S}" not in args.only.split(","):
            continue
        xs = [torch.randn(B, S, S, Cin, device=d).to(dt) for _ in range(args.rotate)]
        ys = [torch.empty(B, S, S, Cout, device=d, dtype=dt) for _ in range(args.rotate)]
        rs = [torch.randn(B, S, S, Cout, device=d).to(dt) for _ in range(args.rotate)] if args.res else None
        w = (torch.randn(Cout, 3, 3, Cin, device=d) / (3 * Cin ** 0.5)).to(dt)
        bias = torch.randn(Cout, device=d)
        st = torch.zeros(B, 16, Cout, 2, device=d)
        geo = dict(B=B, H=S, W=S, R=3, S=3, pad=1, stride=1, Ho=S, Wo=S)
        flops = 2.0 * B * S * S * Cout * 9 * Cin
        i = [0]

        def run():
            i[0] += 1
            k = i[0] % args.rotate
            ops.conv_nt(xs[k], w, ys[k], Cin=Cin, Cout=Cout, ldx=Cin, ldw=9 * Cin, ldy=Cout, bias=bias, stats=st, ldstats=Cout,
                        stats_slots=16, res=rs[k] if rs else None, ldres=Cout if rs else 0, res_scale=0.7, **geo)
        for cfg in cfgs:
            for per in persists:
                _lib.set_tuning("JG_HALO_CFG", cfg)
                if per is not None:
                    _lib.set_tuning("JG_PERSIST64", per)
                line = f"{Cin:5d} {Cout:5d} {S:4d} {cnt:3d} {cfg:3d} {'-' if per is None else per:>3} |"
                for v in dbgs:
                    _lib.set_tuning("JG_HALO_DBG", v)
                    t = timeit(run)
                    line += f" {t * 1e6:7.1f} {flops / t / 1e12:5.0f} |"
                    if v == 0:
                        tot[(cfg, per)] = tot.get((cfg, per), 0.0) + t * cnt
                _lib.set_tuning("JG_HALO_DBG", 0)
                print(line, flush=True)
    for k, v in tot.items():
        print(f"TOTAL cfg={k[0]} persist={k[1]}: {v * 1e3:.2f} ms per step-equivalent")


if __name__ == "__main__":
    main()

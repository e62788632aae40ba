"""Per-layer microbenchmark of the MFMA conv kernels at the BASELINE configs[1] shapes
(DDPM efficient UNet, 256x256, batch 32): forward (conv_nt), input gradient (conv_nt on the
flipped weights) and weight gradient (wgrad_tn), TFLOP/s from HIP events.  Dev tool (GPU box)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import ops
from joligen_amd.ops import JG_OUT_ATOMIC_F32

# (Cin, Cout, k, S, count) -- BASELINE.md conv table (efficient=True), 256x256
SHAPES = [
    (512, 512, 3, 32, 11), (64, 64, 3, 256, 7), (128, 128, 3, 128, 7), (256, 256, 3, 64, 7), (1024, 512, 3, 32, 2),
    (512, 512, 3, 64, 1), (256, 256, 3, 128, 1), (128, 128, 3, 256, 1), (128, 64, 3, 256, 2), (768, 256, 3, 64, 1),
    (384, 128, 3, 128, 1), (192, 64, 3, 256, 1), (512, 256, 3, 64, 1), (256, 128, 3, 128, 1), (768, 512, 3, 32, 1),
    (384, 256, 3, 64, 1), (192, 128, 3, 128, 1), (64, 128, 3, 128, 1), (128, 256, 3, 64, 1), (256, 512, 3, 32, 1),
    (1024, 512, 1, 32, 2), (128, 64, 1, 256, 2), (8, 64, 3, 256, 1), (64, 8, 3, 256, 1),
]


def timeit(fn, reps=5, warm=2):
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
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--target-blocks", type=int, default=0)
    ap.add_argument("--bias", type=int, default=0)
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many operand sets (defeats the 256 MB MALL)")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    if args.target_blocks:
        ops.WGRAD_TARGET_BLOCKS = args.target_blocks
    d = torch.device("cuda:0")
    B = args.batch
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad_v1": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    print(f"{'Cin':>5} {'Cout':>5} k {'S':>4} cnt | {'fwd ms':>8} {'TF':>6} | {'dgrad ms':>8} {'TF':>6} | {'wgv1 ms':>8} {'TF':>6} | {'wgv2 ms':>8} {'TF':>6} splitk")
    for Cin, Cout, k, S, cnt in SHAPES:
        pad = k // 2
        xs = [torch.randn(B, S, S, Cin, device=d).to(dt) for _ in range(args.rotate)]
        ys = [torch.randn(B, S, S, Cout, device=d).to(dt) for _ in range(args.rotate)]
        cnt_box = [0]

        def nxt():
            cnt_box[0] += 1
            return xs[cnt_box[0] % args.rotate], ys[cnt_box[0] % args.rotate]
        x, y = xs[0], ys[0]
        db = torch.zeros(Cout, device=d, dtype=torch.float32) if args.bias else None
        w = (torch.randn(Cout, k, k, Cin, device=d) / (k * Cin ** 0.5)).to(dt)
        wT = (torch.randn(Cin, k, k, Cout, device=d) / (k * Cout ** 0.5)).to(dt)
        dx = torch.empty(B, S, S, Cin, device=d, dtype=dt)
        dw = torch.zeros(Cout, k, k, Cin, device=d, dtype=torch.float32)
        geo = dict(B=B, H=S, W=S, R=k, S=k, pad=pad, stride=1, Ho=S, Wo=S)
        flops = 2.0 * B * S * S * Cout * k * k * Cin
        res = {}
        if "fwd" in args.only or not args.only:
            t = timeit(lambda: ops.conv_nt(x, w, y, Cin=Cin, Cout=Cout, ldx=Cin, ldw=k * k * Cin, ldy=Cout, **geo))
            res["fwd"] = t
        if ("dgrad" in args.only or not args.only) and Cin >= 8:
            t = timeit(lambda: ops.conv_nt(y, wT, dx, Cin=Cout, Cout=Cin, ldx=Cout, ldw=k * k * Cout, ldy=Cin, **geo))
            res["dgrad"] = t
        splitk = 0
        if "wgrad" in args.only or not args.only:
            ktot = k * k * Cin
            tiles = ((Cout + 127) // 128) * ((ktot + 127) // 128)
            splitk = ops._wgrad_splitk(tiles, B * S * S)
            for var in ("3", "2"):
                os.environ["JG_WGRAD_VARIANT"] = var
                def run():
                    xx, yy = nxt()
                    ops.wgrad_tn(yy, xx, dw, Cin=Cin, Cout=Cout, lddy=Cout, ldx=Cin, lddw=ktot, splitk=splitk,
                                 out_mode=JG_OUT_ATOMIC_F32, dbias=db, **geo)
                t = timeit(run)
                res["wgrad" if var == "2" else "wgrad_v1"] = t
        line = f"{Cin:5d} {Cout:5d} {k} {S:4d} {cnt:3d} |"
        for key in ("fwd", "dgrad", "wgrad_v1", "wgrad"):
            if key in res:
                line += f" {res[key] * 1e3:8.3f} {flops / res[key] / 1e12:6.0f} |"
                tot[key][0] += res[key] * cnt
                tot[key][1] += flops * cnt
            else:
                line += f" {'-':>8} {'-':>6} |"
        print(line, splitk, flush=True)
    for key, (t, f) in tot.items():
        if t:
            print(f"TOTAL {key}: {t * 1e3:.2f} ms per step-equivalent, {f / t / 1e12:.0f} TFLOP/s")


if __name__ == "__main__":
    main()

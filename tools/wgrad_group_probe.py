"""Replays a weight-gradient problem list dumped by JG_WGRAD_DUMP=<file> (one flush of ops.deferred_wgrads: the generator backward of a CUT
step) on random operands: every problem alone, then the list as grouped launches under JG_WGRAD_GROUP_BLOCKS settings.
usage: python tools/wgrad_group_probe.py gpurun_out/wgrad_dump.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd import _lib, ops

lines = open(sys.argv[1]).read().split("\n")
n = int(lines[0].split()[1])
probs = [tuple(map(int, l.split())) for l in lines[1:1 + n]]
dev = torch.device("cuda", 0)
dt = torch.bfloat16
ops_ = []
for (B, H, W, Cin, Cout, R, S, pad, stride, Ho, Wo, splitk, hasb) in probs:
    x = torch.randn(B, H, W, Cin, device=dev, dtype=dt)
    dy = torch.randn(B, Ho, Wo, Cout, device=dev, dtype=dt)
    dw = torch.zeros(Cout, R * S * Cin, device=dev, dtype=torch.float32)
    db = torch.zeros(Cout, device=dev, dtype=torch.float32) if hasb else None
    ops_.append((dy, x, dw, db, dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=S, pad=pad, stride=stride, Ho=Ho, Wo=Wo, lddy=Cout, ldx=Cin, lddw=R * S * Cin, splitk=splitk)))


def time_fn(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = 0.0
rows = []
for dy, x, dw, db, kw in ops_:
    t = time_fn(lambda: ops.wgrad_tn(dy, x, dw, dbias=db, **kw))
    byts = 2.0 * (dy.numel() + x.numel())
    rows.append((t, kw, byts))
    tot += t
rows.sort(key=lambda r: -r[0])
print("alone: sum %.1f us over %d problems" % (tot, len(rows)))
for t, kw, byts in rows[:25]:
    print("  %7.1f us  %5.2f TB/s  B %d H %d Cin %d Cout %d R %d stride %d splitk %d" % (t, byts / t / 1e6, kw["B"], kw["H"], kw["Cin"], kw["Cout"], kw["R"], kw["stride"], kw["splitk"]))


def grouped():
    with ops.deferred_wgrads():
        for dy, x, dw, db, kw in ops_:
            ops.wgrad_tn(dy, x, dw, dbias=db, **kw)


for gb in (0, 512, 1024, 2048):
    _lib.set_tuning("JG_WGRAD_GROUP_BLOCKS", gb)
    print("grouped, JG_WGRAD_GROUP_BLOCKS=%d: %.1f us" % (gb, time_fn(grouped)))

# per group of the C side's grouping (problems of one tile class in arrival order, 16 per launch; halo-served 3x3 / 7x7 problems leave singly)
_lib.set_tuning("JG_WGRAD_GROUP_BLOCKS", 1024)
special = lambda kw: kw["R"] > 1 and kw["stride"] == 1 and kw["R"] in (3, 7)
for cls in (1, 2):
    sel = [o for o in ops_ if not special(o[4]) and (1 if o[4]["Cout"] <= 64 else 2) == cls]
    for g0 in range(0, len(sel), 16):
        grp = sel[g0:g0 + 16]

        def run():
            with ops.deferred_wgrads():
                for dy, x, dw, db, kw in grp:
                    ops.wgrad_tn(dy, x, dw, dbias=db, **kw)

        t = time_fn(run)
        byts = sum(2.0 * (dy.numel() + x.numel()) for dy, x, dw, db, kw in grp)
        flops = sum(2.0 * kw["B"] * kw["Ho"] * kw["Wo"] * kw["Cout"] * kw["Cin"] * kw["R"] * kw["S"] for dy, x, dw, db, kw in grp)
        print("class %d group %d: %7.1f us, %6.1f MB operands (%.2f TB/s), %.1f GFLOP (%.0f TFLOP/s)" % (cls, g0 // 16, t, byts / 1e6, byts / t / 1e6, flops / 1e9, flops / t / 1e6))
        for dy, x, dw, db, kw in grp:
            print("      B %d H %d Cin %d Cout %d R %d stride %d" % (kw["B"], kw["H"], kw["Cin"], kw["Cout"], kw["R"], kw["stride"]))
for o in ops_:
    if special(o[4]):
        dy, x, dw, db, kw = o
        print("special: %.1f us  B %d H %d Cin %d Cout %d R %d" % (time_fn(lambda: ops.wgrad_tn(dy, x, dw, dbias=db, **kw)), kw["B"], kw["H"], kw["Cin"], kw["Cout"], kw["R"]))

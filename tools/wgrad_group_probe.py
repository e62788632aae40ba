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

"""dev probe: single-pass GroupNorm backward (csrc/gn_fused.hip) against reduce -> coef -> apply on the UNet's shapes (batch 32)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from joligen_amd import _lib, ops
from joligen_amd.ops import _st

L = _lib.lib()
dev = torch.device("cuda:0")
BF = _lib.JG_BF16
junk = torch.empty(600 << 20, dtype=torch.uint8, device=dev)


def run(B, H, W, C, variants, up=False, reps=6, act=1, G=32):
    HW = H * W
    x = torch.randn(B, HW, C, device=dev).bfloat16()
    dy = torch.randn(B, (HW // 4) if up else HW, C, device=dev).bfloat16()
    dx = torch.empty_like(x)
    ab = torch.randn(B, C, 2, device=dev)
    mr = torch.rand(B, G, 2, device=dev) + 0.5
    red = torch.zeros(B, C, 2, device=dev)
    cnt = torch.zeros(B, 2, device=dev, dtype=torch.int32)
    pqr = torch.empty(B, C, 3, device=dev)
    st = _st()
    status = ops.gn_status(dev)

    def old():
        red.zero_()
        if up:
            L.jg_gn_bwd_reduce_up_acc(BF, x.data_ptr(), C, dy.data_ptr(), C, 0.25, ab.data_ptr(), red.data_ptr(), B, H, W, C, act, st)
        else:
            L.jg_gn_bwd_reduce_ld_acc(BF, x.data_ptr(), C, dy.data_ptr(), C, ab.data_ptr(), red.data_ptr(), B, HW, C, act, st)
        L.jg_gn_bwd_coef_slots(red.data_ptr(), 1, None, None, None, 0, mr.data_ptr(), pqr.data_ptr(), None, None, None, 0, B, HW, C, G, st)
        if up:
            L.jg_gn_bwd_apply_up(BF, x.data_ptr(), C, dy.data_ptr(), C, 0.25, ab.data_ptr(), pqr.data_ptr(), dx.data_ptr(), C, None, 0, 0.0, None, 0, 0.0, B, H, W, C, act, st)
        else:
            L.jg_gn_bwd_apply_ld(BF, x.data_ptr(), C, dy.data_ptr(), C, ab.data_ptr(), pqr.data_ptr(), dx.data_ptr(), C, None, 0, 0.0, None, 0, 0.0, B, HW, C, act, st)

    def fused():
        red.zero_()
        cnt.zero_()
        rc = L.jg_gn_bwd_fused(BF, int(up), x.data_ptr(), C, dy.data_ptr(), C, 0.25 if up else 1.0, ab.data_ptr(), red.data_ptr(), cnt.data_ptr(),
                               status.data_ptr(), None, None, None, 0, mr.data_ptr(), None, None, None, 0, G, dx.data_ptr(), C, None, 0, 0.0, None, 0, 0.0,
                               B, H, W, C, act, st)
        assert rc == 0, rc

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        tot = 0.0
        for _ in range(reps):
            junk.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps * 1e3

    t_old = timeit(old)
    ref = dx.float().clone()
    nbytes = x.numel() * 2 * (2 + (0.25 if up else 1))
    line = f"B={B} {H}x{W} C={C} up={int(up)}: 3-launch {t_old:7.1f} us ({nbytes / t_old / 1e6:5.2f} TB/s on 3N)"
    for depth, cap in variants:
        _lib.set_tuning("JG_GN_FUSED", depth)
        _lib.set_tuning("JG_GN_FUSED_CAP", cap)
        t = timeit(fused)
        err = float((dx.float() - ref).norm() / ref.norm())
        line += f" | N{depth}/cap{cap} {t:7.1f} ({nbytes / t / 1e6:4.2f}) e={err:.1e}"
    print(line, flush=True)
    assert int(status[0]) == 0, "spin expired"


def abl(shape, depth=16, cap=256, up=False):
    for dbg, sleep in ((0, 4), (0, 1), (0, 16), (1, 4), (2, 4), (3, 4), (4, 4), (7, 4)):
        _lib.set_tuning("JG_GN_FUSED_DBG", dbg)
        _lib.set_tuning("JG_GN_FUSED_SLEEP", sleep)
        print(f"dbg={dbg} sleep={sleep}: ", end="")
        run(*shape, [(depth, cap)], up=up, reps=4)
    _lib.set_tuning("JG_GN_FUSED_DBG", 0)
    _lib.set_tuning("JG_GN_FUSED_SLEEP", 4)


if len(sys.argv) > 1 and sys.argv[1] == "abl":
    abl((32, 256, 256, 64))
    abl((32, 256, 256, 128))
    abl((32, 128, 128, 128))
    sys.exit(0)
VAR = [(8, 256), (12, 256), (16, 256), (20, 256)]
for shape in ((32, 256, 256, 64), (32, 256, 256, 128), (32, 256, 256, 192), (32, 128, 128, 128), (32, 128, 128, 256), (32, 128, 128, 384),
              (32, 64, 64, 256), (32, 64, 64, 512), (32, 64, 64, 768), (32, 32, 32, 512), (32, 32, 32, 1024)):
    run(*shape, VAR)
run(32, 256, 256, 64, VAR, up=True)
run(32, 128, 128, 128, VAR, up=True)

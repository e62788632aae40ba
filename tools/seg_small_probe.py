"""Device timing of the small SegFormer-generator backward kernels at the shapes of the CUT configs[2] step (MiT-B0 on 2 x 16
images of 256^2): depth-wise 3x3 weight gradient (atomics from every block vs workspace + summing launch, and the grid cap) and the
LayerNorm backward under different grid caps (its dgamma / dbeta atomics are a same-address chain as deep as the grid).

usage (GPU box): python tools/seg_small_probe.py > gpurun_out/seg_small_probe.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joligen_amd import _lib, ops_segformer as S  # noqa: E402
from joligen_amd._lib import check  # noqa: E402

D = "cuda:0"
REPS = 200


def timed(fn):
    """20 calls captured in a HIP graph, replayed: no host time between the kernels"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS // 20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / (REPS // 20 * 20)


def param(shape):
    p = torch.nn.Parameter(torch.randn(shape, device=D))
    p.grad = torch.zeros_like(p)
    return p


def dw_case(B, H, W, C, dtype):
    """the backward entry point alone (ctypes, no autograd): device time is what the events see"""
    x, pre, gy = (torch.randn(B, H, W, C, device=D, dtype=dtype) for _ in range(3))
    du, dx = torch.empty_like(x), torch.empty_like(x)
    w, b = param((C, 1, 3, 3)), param((C,))
    lib = _lib.lib()

    def run():
        nws = int(lib.jg_dwconv3x3_bwd_ws_floats(B, H, W, C)) if S.DW_TWO_PHASE else 0
        ws = torch.empty(max(nws, 1), device=D, dtype=torch.float32)
        check(lib.jg_dwconv3x3_bwd_ws(S._dt(x), x.data_ptr(), pre.data_ptr(), gy.data_ptr(), w.data_ptr(), du.data_ptr(), None,
                                      w.grad.data_ptr(), b.grad.data_ptr(), ws.data_ptr() if nws else None, nws, B, H, W, C, 1, S._st()), "dw")
    return run


def ln_case(R, C, dtype):
    x, gy = torch.randn(R, C, device=D, dtype=dtype), torch.randn(R, C, device=D, dtype=dtype)
    dx = torch.empty_like(x)
    mr = torch.stack([torch.zeros(R, device=D), torch.ones(R, device=D)], 1).contiguous()
    w, b = param((C,)), param((C,))
    lib = _lib.lib()

    def run():
        check(lib.jg_layernorm_bwd(S._dt(x), x.data_ptr(), gy.data_ptr(), w.data_ptr(), mr.data_ptr(), dx.data_ptr(), w.grad.data_ptr(),
                                   b.grad.data_ptr(), R, C, S._st()), "ln")
    return run


def main():
    dtype = torch.bfloat16
    lib = _lib.lib()
    print("backward entry point of one op through ctypes, us per call (torch events over %d calls replayed from a HIP graph)" % REPS)
    print("\ndepth-wise 3x3 + GELU, B=32: weight-gradient launch(es) only (dx = NULL)")
    for (H, C) in [(64, 128), (32, 256), (16, 640), (8, 1024)]:
        row = []
        for two, cap, ppt in [(False, 1024, 8), (True, 1024, 8), (True, 512, 8), (True, 1024, 4), (True, 512, 4), (True, 1024, 2), (True, 2048, 2), (True, 2048, 1)]:
            S.DW_TWO_PHASE = two
            lib.jg_set_tuning(b"JG_DW_BWD_CAP", cap)
            lib.jg_set_tuning(b"JG_DW_BWD_PPT", ppt)
            row.append("%s cap %4d ppt %d: %6.1f" % ("2ph" if two else "atm", cap, ppt, timed(dw_case(32, H, H, C, dtype))))
        print("  %3dx%-3d C=%-4d | %s" % (H, H, C, " | ".join(row)))
    S.DW_TWO_PHASE = True
    lib.jg_set_tuning(b"JG_DW_BWD_CAP", 1024)
    lib.jg_set_tuning(b"JG_DW_BWD_PPT", 8)
    print("\nLayerNorm, rows x channels")
    for (R, C) in [(32 * 4096, 32), (32 * 1024, 64), (32 * 256, 160), (32 * 64, 256), (32 * 64, 32)]:
        row = []
        for cap in [1024, 256, 128, 64, 32]:
            lib.jg_set_tuning(b"JG_LN_BWD_CAP", cap)
            row.append("cap %4d: %6.1f" % (cap, timed(ln_case(R, C, dtype))))
        print("  %7d x %-4d | %s" % (R, C, " | ".join(row)))
    lib.jg_set_tuning(b"JG_LN_BWD_CAP", 256)


if __name__ == "__main__":
    main()

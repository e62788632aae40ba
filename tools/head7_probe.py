"""Timing of the row-packed 7x7 head (ops.head_conv7) against the 7x7 kernels of round 5 at the CUT shape (B 32, 256^2, 64 -> 3, + Tanh):
forward, forward + backward; prints the kernel instances the dispatch reported.  usage: python tools/head7_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from joligen_amd import _lib, ops
from joligen_amd.arena import ParamArena
from joligen_amd.modules.layers import JGConv2d

dev = torch.device("cuda", 0)
dt = torch.bfloat16


class M(nn.Module):
    def __init__(self):
        super().__init__()
        self.c = JGConv2d(64, 3, 7, padding=0)


m = M()
ParamArena(m, dev, dt, priority=()).refresh()
B, H, W = 32, 256, 256
x = torch.randn(B, H + 6, W + 6, 64, device=dev, dtype=dt).requires_grad_(True)
g = torch.randn(B, H, W, 8, device=dev, dtype=dt)


def t(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def packed(bwd):
    y = ops.head_conv7(x, m.c.meta, 4)
    if bwd:
        y.backward(g)


def old(bwd):
    y = ops.activation(m.c(x), 4)
    if bwd:
        y.backward(g)


ops.KERNEL_TIMING = None
print("packed fwd %.1f us, fwd+bwd %.1f us" % (t(lambda: packed(False)), t(lambda: packed(True))))
print("7x7    fwd %.1f us, fwd+bwd %.1f us" % (t(lambda: old(False)), t(lambda: old(True))))
z = torch.empty(B, H + 6, W, 32, device=dev, dtype=dt)
wz = torch.zeros(32, 7, 64, device=dev, dtype=dt)
ops.conv_nt(x.detach(), wz, z, B=B, H=H + 6, W=W + 6, Cin=64, Cout=32, R=1, S=7, pad=0, stride=1, Ho=H + 6, Wo=W, ldx=64, ldw=7 * 64, ldy=32)
print("forward stage ran on:", _lib.lib().jg_last_kernel().decode())
print("1x7 stage alone %.1f us" % t(lambda: ops.conv_nt(x.detach(), wz, z, B=B, H=H + 6, W=W + 6, Cin=64, Cout=32, R=1, S=7, pad=0, stride=1, Ho=H + 6, Wo=W, ldx=64, ldw=7 * 64, ldy=32)))
# the pieces of the packed backward
Hp, Wp, Hz = H + 6, W + 6, (H + 6 + 7) // 8 * 8
out = torch.randn(B, H, W, 8, device=dev, dtype=dt).tanh()
dz = torch.empty(B, Hz, W, 32, device=dev, dtype=dt)
dzm = torch.empty(B, Hp, W + 12, 32, device=dev, dtype=dt)
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
print("tapspread7 %.1f us" % t(lambda: L.jg_tapspread7(1 if dt == torch.bfloat16 else 0, g.data_ptr(), out.data_ptr(), dz.data_ptr(), dzm.data_ptr(), B, H, W, Hz, 4, st())))
wzT = torch.zeros(64, 7, 32, device=dev, dtype=dt)
dx = torch.empty(B, Hp, Wp, 64, device=dev, dtype=dt)
f = lambda: ops.conv_nt(dzm, wzT, dx, B=B, H=Hp, W=W + 12, Cin=32, Cout=64, R=1, S=7, pad=0, stride=1, Ho=Hp, Wo=Wp, ldx=32, ldw=7 * 32, ldy=64)
f()
print("input-gradient stage ran on:", L.jg_last_kernel().decode(), "%.1f us" % t(f))
dwz = torch.zeros(32, 7, 64, device=dev, dtype=torch.float32)
db = torch.zeros(32, device=dev, dtype=torch.float32)
f = lambda: ops.wgrad_tn(dz, x.detach(), dwz, B=B, H=Hp, W=Wp, Cin=64, Cout=32, R=1, S=7, pad=0, stride=1, Ho=Hz, Wo=W, lddy=32, ldx=64, lddw=7 * 64, dbias=db, Cin_out=64, Cout_out=32, splitk=1, defer=False)
f()
print("weight-gradient stage ran on:", L.jg_last_kernel().decode(), "%.1f us" % t(f))
z = torch.empty(B, Hp, W, 32, device=dev, dtype=dt)
o2 = torch.empty(B, H, W, 8, device=dev, dtype=dt)
print("tapsum7 %.1f us" % t(lambda: L.jg_tapsum7(1 if dt == torch.bfloat16 else 0, z.data_ptr(), None, o2.data_ptr(), B, H, W, 4, st())))

"""Fused forward/backward schedule of the DDPM UNet (the training hot path of SURVEY.md 8 a10-a14).

`UNet.forward` (unet_generator_attn.py here) mirrors the reference module by module on top of
per-op autograd Functions.  This executor runs the SAME modules and the SAME kernels as ONE
autograd node with a hand-written backward, which is what lets the step shed its HBM-bound glue:

  * GroupNorm statistics come out of the PRODUCING convolution's epilogue (fp32 atomics on
    (image, channel) sums) -- the separate statistics pass over every activation is gone;
  * `torch.cat([h, hs.pop()], 1)` (reference unet_generator_attn.py:692-693) is never executed: both
    producers store straight into the halves of a pre-allocated concat buffer (strided epilogue
    stores) and every consumer kernel takes explicit pixel strides; the backward split is a view;
  * every gradient fan-in (ResBlock skip, attention residual, UNet skip connections) is folded into
    the GroupNorm-backward pass of the consumer (`jg_gn_bwd_apply_ld` addends) or into the epilogue
    of the 1x1 skip-convolution's input-gradient, instead of autograd's add kernels.

Parameter gradients are accumulated by the kernels straight into the arena (`param.grad` views).
"""
from __future__ import annotations

import contextlib
import math
import os
import weakref
from typing import Tuple

import torch
import torch.nn as nn

from .._autograd import JGFunction
from .. import _lib, ops, parallel
from .._lib import JG_ACT_NONE, JG_ACT_SILU, check
from ..ops import _dt, _p, _st, attn_core_bwd, attn_core_fwd, conv_nt, wgrad_tn


class Act:
    """An activation of the schedule: NHWC view `t` (any pixel stride), fp32 statistics view
    `st` [B, NSLOT, C, 2] (sum, sum^2 per image and channel, NSLOT partial replicas, any row stride) taken over `hw` pixels."""

    __slots__ = ("t", "st", "hw", "pid", "hs_j")

    def __init__(self, t, st, hw, pid, hs_j=None):
        self.t, self.st, self.hw, self.pid, self.hs_j = t, st, hw, pid, hs_j


def _ld(t):
    assert t.stride(-1) == 1
    return t.stride(-2)


def _stats_fusable(B, Ho, Wo, Cout):
    return (Ho * Wo) % 256 == 0 and Cout % 64 == 0


# replicas of every statistics row: a conv tile adds into replica (tile index % NSLOT), which keeps the
# same-address fp32 atomic chains of the full-resolution layers short (256 tiles per image at 256x256)
NSLOT = 16
# GroupNorm-backward reductions inside the producing dgrad convolution (jg_conv_args.stats_mode 1).  Measured on
# MI355X (profiles/r01_notes.md): the separate reduction pass disappears (-5.4 ms/step) but the un-overlapped epilogue
# reads of x cost the convolutions +4.8 ms and the slot-summing bwd_coef +1.8 ms: OFF by default.
NORM_BEFORE_UP = os.environ.get("JG_NORM_BEFORE_UP", "1") != "0"
# ResBlock-down: pool(act(norm(x))) in one kernel, and the GroupNorm backward reads the pooled-resolution gradients through the
# upsample index map (jg_gn_apply_pool / jg_gn_bwd_*_up) instead of materialising full-resolution copies
FUSE_DOWN_POOL = os.environ.get("JG_FUSE_DOWN_POOL", "1") != "0"
# forward of the up-block conv2 in its sub-pixel form (x_mode 2: 16 instead of 36 tap-MACs per low-resolution pixel)
SUBPIXEL_CONV = os.environ.get("JG_SUBPIXEL_CONV", "1") != "0"
POOL_IN_DGRAD = os.environ.get("JG_POOL_IN_DGRAD", "1") != "0"
X_UP_ON_READ = os.environ.get("JG_X_UP_ON_READ", "1") != "0"
RES_UP_ON_READ = os.environ.get("JG_RES_UP_ON_READ", "1") != "0"
FUSE_GN_REDUCE = os.environ.get("JG_FUSE_GN_REDUCE", "0") != "0"
# Weight gradients on a SECOND HIP stream.  The backward's critical chain is dgrad -> GroupNorm backward -> dgrad -> ...; the weight
# gradient of a layer hangs off it (nothing in the chain reads it).  Launched on their own stream the MFMA-bound weight-gradient
# kernels run underneath the HBM-bound GroupNorm-backward passes of the chain instead of in front of them.  The streams meet again at
# the end of the backward (and before a gradient chunk leaves for the all-reduce).
WGRAD_STREAM = os.environ.get("JG_WGRAD_STREAM", "1") != "0"
# JG_WGRAD_CU_MASK = "<first>:<count>" (A/B): the weight-gradient stream is created with hipExtStreamCreateWithCUMask over CU-mask bits
# [first, first + count) of the 256 (bit i = compute unit i / 8 of XCD i % 8): a fixed CU partition for the MFMA-bound weight gradients
# instead of free competition with the HBM-bound normalisation passes of the chain.  Unset = a plain stream.
WGRAD_CU_MASK = os.environ.get("JG_WGRAD_CU_MASK", "")


def cu_masked_stream(device, first, count, total=256):
    """a torch stream over `hipExtStreamCreateWithCUMask` (the C ABI takes any hipStream_t: kernels launched on it run on the masked CUs)"""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    words = (total + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(first, min(total, first + count)):
        mask[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with {rc}")
    return torch.cuda.ExternalStream(st.value, device=device)


def _side_stream(device):
    if WGRAD_CU_MASK:
        first, count = (int(v) for v in WGRAD_CU_MASK.split(":"))
        return cu_masked_stream(device, first, count)
    return torch.cuda.Stream(device=device)
# issue order of a 3x3 layer's two backward convolutions: 1 = weight gradient first (it starts together with the input gradient and has
# that kernel's time plus the GroupNorm-backward passes behind it to finish in); 0 = input gradient first (the side stream then waits
# for it as well).  Same-box A/B, two runs each: 49.19 vs 49.49 ms/step (profiles/r04_wgrad_first_ab.log).
WGRAD_FIRST = os.environ.get("JG_WGRAD_FIRST", "1") != "0"
# GroupNorm backward: coefficient step inside the apply pass (jg_gn_bwd_apply_fc, 57 launches per step fewer).  Measured 51.9 vs 51.6 ms
# (A/B on one box): the ~50 us a 6 us coefficient kernel spends waiting next to the weight-gradient stream is paid by the next kernel
# instead, and the per-workgroup prologue costs what the launch saved -- off by default, kept for single-stream configurations.
FUSE_GN_COEF = os.environ.get("JG_FUSE_GN_COEF", "0") != "0"
# GroupNorm backward as ONE launch with x / dy resident in registers between the reduction and the apply step (csrc/gn_fused.hip,
# DESIGN.md 4.5b): 6 instead of 10 bytes per element on paper.  MEASURED SLOWER on every UNet shape (round 4: 399 vs 310 us at 64 ch x
# 256^2 x 32: the wait for the image's other workgroups idles the CU slots that hold the data): off; 1 = use it (kept, tested).
GN_FUSED = os.environ.get("JG_GN_FUSED", "0") != "0"
# ResBlocks whose channel count changes read their input twice (act(norm(x)) and skip_connection(x)): one launch does both
# (jg_conv1x1_gn_apply, the streaming 1x1 kernel also writes the normalised activation).  0 = gn_apply + 1x1 convolution.
FUSE_SKIP_APPLY = os.environ.get("JG_FUSE_SKIP_APPLY", "1") != "0"
# ... and in the backward the input gradient of that skip convolution carries the GroupNorm-backward apply step of the same tensor in its
# epilogue (jg_conv1x1_gn_bwd_apply): the [M][Cin] intermediate between the two launches is neither written nor read.
FUSE_SKIP_BWD = os.environ.get("JG_FUSE_SKIP_BWD", "1") != "0"


class _Pool:
    """Zero-initialised fp32 scratch for the statistics rows of one forward (one fill kernel)."""

    def __init__(self, n, device):
        self.buf = torch.zeros(n, device=device, dtype=torch.float32)
        self.off = 0

    def take(self, B, C):
        n = B * NSLOT * C * 2
        if self.off + n > self.buf.numel():
            raise RuntimeError("statistics pool exhausted")
        v = self.buf[self.off:self.off + n].view(B, NSLOT, C, 2)
        self.off += n
        return v

    def take_rows(self, B, C):
        """[B, C, 2] zeroed rows (the reductions of a stand-alone GroupNorm backward)"""
        n = B * C * 2
        if self.off + n > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + n].view(B, C, 2)
        self.off += n
        return v


# ---------------------------------------------------------------------------------------------------
# raw launches (no autograd): every tensor may be a channel slice of a wider buffer
# ---------------------------------------------------------------------------------------------------
def halo_ok(m, H, W):
    """shape limits of the halo-resident 3x3 kernels (conv_halo.hip / wgrad_halo.hip): the only ones with x_mode / pad_mode"""
    return m.R == 3 and m.S == 3 and m.pad == 1 and m.stride == 1 and m.Cin % 64 == 0 and m.Cout % 64 == 0 and H % 16 == 0 and W % 16 == 0


def subpixel_ok(m, H, W):
    """shape limits of jg_conv_args.x_mode 2 (H, W: the upsampled size)"""
    return halo_ok(m, H, W) and H % 32 == 0 and W % 32 == 0 and m.Cin_real == m.Cin and m.Cout_real == m.Cout


def subpixel_fold(m, dtype):
    """folded weights [4][Cout][2][2][Cin] of the sub-pixel form from the fp32 master weights (physical [Cout][3][3][Cin])"""
    out = torch.empty((4, m.Cout, 2, 2, m.Cin), device=m.weight.device, dtype=dtype)
    check(_lib.lib().jg_subpixel_fold(_lib.JG_F16 if dtype == torch.float16 else _lib.JG_BF16, m.weight.data_ptr(), out.data_ptr(), m.Cout,
                                      m.Cin, _st()), "jg_subpixel_fold")
    return out


def conv_fwd(x, m, out=None, res=None, res_scale=1.0, alpha=1.0, stats=None, res_up=False, x_up=False, wfold=None):
    B, H, W, Cin = x.shape
    if x_up:      # x is the half-resolution tensor, the convolution runs over its nearest upsample (jg_conv_args.x_mode 1)
        H, W = 2 * H, 2 * W
    Ho, Wo = m.out_hw(H, W)
    if out is None:
        out = torch.empty((B, Ho, Wo, m.Cout), device=x.device, dtype=x.dtype)
    fuse = stats is not None and _stats_fusable(B, Ho, Wo, m.Cout)
    # wfold: x_mode 2, the sub-pixel form of the same convolution (four 2x2-tap phases on the folded weights)
    conv_nt(x, m.w16 if wfold is None else wfold, out, B=B, H=H, W=W, Cin=Cin, Cout=m.Cout, R=m.R, S=m.S, pad=m.pad, stride=m.stride,
            Ho=Ho, Wo=Wo, ldx=_ld(x), ldw=m.R * m.S * Cin if wfold is None else 4 * Cin, ldy=_ld(out),
            bias=m.bias_pad if m.bias_pad is not None else m.bias, res=res,
            ldres=_ld(res) if res is not None else 0, alpha=alpha, res_scale=res_scale,
            stats=stats if fuse else None, ldstats=stats.stride(1) // 2 if fuse else 0, stats_slots=NSLOT,
            res_mode=1 if res_up else 0, x_mode=(2 if wfold is not None else 1) if x_up else 0)
    if stats is not None and not fuse:   # shapes the fused epilogue does not cover: separate statistics pass
        check(_lib.lib().jg_gn_stats_ld(_dt(out), out.data_ptr(), _ld(out), stats.data_ptr(), stats.stride(0) // 2, B, Ho * Wo,  # replica 0
                                        m.Cout, _st()), "jg_gn_stats_ld")
    return out


def conv1x1_gn_apply(x, m, ab, act):
    """(act(a x + b), conv1x1(x)) in one launch, or None when the layer is not a streaming-kernel shape"""
    B, H, W, Cin = x.shape
    if m.R != 1 or m.S != 1 or m.stride != 1 or m.pad != 0:
        return None
    y = torch.empty((B, H, W, m.Cout), device=x.device, dtype=x.dtype)
    yn = torch.empty((B, H, W, Cin), device=x.device, dtype=x.dtype)
    ok = conv_nt(x, m.w16, y, B=B, H=H, W=W, Cin=Cin, Cout=m.Cout, R=1, S=1, pad=0, stride=1, Ho=H, Wo=W, ldx=_ld(x), ldw=Cin, ldy=m.Cout,
                 bias=m.bias_pad if m.bias_pad is not None else m.bias, apply=(ab, yn, Cin, act))
    return (yn, y) if ok is not False else None


def conv_dgrad(dy, m, x_shape, out=None, res=None, alpha=1.0, gn=None, pool=None, pool_out=False):
    """Input gradient of a stride-1 convolution.  `gn = (x, ab, act)`: the result is the output-gradient of a
    GroupNorm with input x / coefficients ab; its backward reductions (sum du, sum du*x) are then taken in
    this epilogue and returned as `red` [B, NSLOT, C, 2] (None when the shape is not covered: the caller
    runs the separate reduction pass)."""
    if m.stride != 1:
        raise NotImplementedError("input-gradient of strided convolutions is not implemented")
    B, H, W, Cin = x_shape
    _, Ho, Wo, Cout = dy.shape
    if pool_out:   # the result is the 2x2 sum-pool of the input gradient (jg_conv_args.y_mode 1): adjoint of a conv over Upsample(h)
        assert out is None and res is None and gn is None
        out = torch.empty((B, H // 2, W // 2, Cin), device=dy.device, dtype=dy.dtype)
    if out is None:
        out = torch.empty(x_shape, device=dy.device, dtype=dy.dtype)
    red = None
    kw = {}
    if gn is not None and pool is not None and res is None and _stats_fusable(B, H, W, Cin) and FUSE_GN_REDUCE:
        gx, gab, gact = gn
        red = pool.take(B, Cin)
        kw = dict(stats=red, ldstats=Cin, stats_slots=NSLOT, gn_reduce=(gx, _ld(gx), gab, gact))
    conv_nt(dy, m.w16T, out, B=B, H=Ho, W=Wo, Cin=Cout, Cout=Cin, R=m.R, S=m.S, pad=m.R - 1 - m.pad, stride=1, Ho=H, Wo=W,
            ldx=_ld(dy), ldw=m.R * m.S * Cout, ldy=_ld(out), alpha=alpha, res=res, ldres=_ld(res) if res is not None else 0,
            res_scale=1.0, y_mode=1 if pool_out else 0, **kw)
    return (out, red) if gn is not None else out


def conv_wgrad(dy, x, m, alpha=1.0, dbias_scale=0.0, x_up=False):
    B, H, W, Cin = x.shape
    if x_up:
        H, W = 2 * H, 2 * W
    _, Ho, Wo, Cout = dy.shape
    wg = m.weight.grad
    if wg is None:
        raise RuntimeError("conv weight has no arena-backed .grad")
    ktot = m.R * m.S * Cin
    tiles = ((Cout + 127) // 128) * ((ktot + 127) // 128)
    splitk = ops._wgrad_splitk(tiles, B * Ho * Wo)
    wgrad_tn(dy, x, wg, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=m.R, S=m.S, pad=m.pad, stride=m.stride, Ho=Ho, Wo=Wo,
             lddy=_ld(dy), ldx=_ld(x), lddw=m.R * m.S * m.Cin_real, dbias=m.bias.grad if m.bias is not None else None,
             Cin_out=m.Cin_real, Cout_out=m.Cout_real, splitk=splitk, alpha=alpha, dbias_scale=dbias_scale, x_mode=1 if x_up else 0)


def gn_coef(st, hw, gamma, beta, film, G, eps):
    B, ns, C, _ = st.shape
    ab = torch.empty((B, C, 2), device=st.device, dtype=torch.float32)
    mr = torch.empty((B, G, 2), device=st.device, dtype=torch.float32)
    check(_lib.lib().jg_gn_coef_ld(st.data_ptr(), st.stride(1) // 2, ns, _p(gamma), _p(beta), _p(film),
                                   film.stride(0) if film is not None else 0, ab.data_ptr(), mr.data_ptr(), B, hw, C, G, eps,
                                   _st()), "jg_gn_coef_ld")
    return ab, mr


def gn_apply(x, ab, act):
    B, H, W, C = x.shape
    y = torch.empty((B, H, W, C), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_gn_apply_ld(_dt(x), x.data_ptr(), _ld(x), ab.data_ptr(), y.data_ptr(), C, B, H * W, C, act, _st()),
          "jg_gn_apply_ld")
    return y


def gn_bwd(x, dy, ab, mr, gamma, beta, film, G, act, dfilm=None, out=None, adds=(), red=None, pooled=None, pool=None, skip=None):
    """dx = GroupNorm-backward(x, dy) + sum_i scale_i * add_i  (at most two addends, fused).
    `red`: reductions already accumulated by the convolution that produced dy ([B, NSLOT, C, 2]).
    `pooled`: (dy_scale, low_add): dy (and the optional addend `low_add` = (tensor, scale)) live at the 2x2-POOLED resolution; the
    kernels read them through the nearest-upsample map (adjoint of the average pool) instead of materialising the upsampled copy.
    `skip` = (dO, m, alpha): x also feeds the 1x1 convolution m (the ResBlock's skip_connection) whose output gradient is dO: the result is
    alpha * dgrad_m(dO) + GroupNorm-backward(x, dy) + addends, the apply step running in the epilogue of the input-gradient launch."""
    L = _lib.lib()
    B, H, W, C = x.shape
    HW = H * W
    dev, dt = x.device, _dt(x)
    nslots = NSLOT
    if red is None and GN_FUSED and skip is None and pool is not None and G <= 2048 and C <= 2048:
        rows, cnt = pool.take_rows(B, C), pool.take_rows(B, 1)
        if rows is not None and cnt is not None:
            dgamma = gamma.grad if gamma is not None else None
            dbeta = beta.grad if beta is not None else None
            if gamma is not None and dgamma is None:
                raise RuntimeError("norm weight has no arena-backed .grad")
            if out is None:
                out = torch.empty((B, H, W, C), device=dev, dtype=x.dtype)
            adds = list(adds)
            if pooled is not None:
                if len(adds) > 1:
                    raise RuntimeError("at most one full-resolution addend next to the pooled one")
                a1, s1 = pooled[1] if pooled[1] is not None else (None, 0.0)
                a2, s2 = adds[0] if adds else (None, 0.0)
            else:
                if len(adds) > 2:
                    raise RuntimeError("at most two fused gradient addends")
                a1, s1 = adds[0] if len(adds) > 0 else (None, 0.0)
                a2, s2 = adds[1] if len(adds) > 1 else (None, 0.0)
            check(L.jg_gn_bwd_fused(dt, int(pooled is not None), x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy),
                                    pooled[0] if pooled is not None else 1.0, ab.data_ptr(), rows.data_ptr(), cnt.data_ptr(),
                                    ops.gn_status(dev).data_ptr(), _p(gamma), _p(beta), _p(film), film.stride(0) if film is not None else 0,
                                    mr.data_ptr(), _p(dgamma), _p(dbeta), _p(dfilm), dfilm.stride(0) if dfilm is not None else 0, G,
                                    out.data_ptr(), _ld(out), _p(a1), _ld(a1) if a1 is not None else 0, s1, _p(a2),
                                    _ld(a2) if a2 is not None else 0, s2, B, H, W, C, act, _st()), "jg_gn_bwd_fused")
            return out
    if red is None:
        nslots = 1
        red = pool.take_rows(B, C) if pool is not None else None      # zeroed once per backward pass: no memset launch per layer
        up, ld = (L.jg_gn_bwd_reduce_up_acc, L.jg_gn_bwd_reduce_ld_acc) if red is not None else (L.jg_gn_bwd_reduce_up, L.jg_gn_bwd_reduce_ld)
        if red is None:
            red = torch.empty((B, C, 2), device=dev, dtype=torch.float32)
        if pooled is not None:
            check(up(dt, x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), pooled[0], ab.data_ptr(), red.data_ptr(), B, H, W, C, act, _st()),
                  "jg_gn_bwd_reduce_up")
        else:
            check(ld(dt, x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), ab.data_ptr(), red.data_ptr(), B, HW, C, act, _st()),
                  "jg_gn_bwd_reduce_ld")
    dgamma = gamma.grad if gamma is not None else None
    dbeta = beta.grad if beta is not None else None
    if gamma is not None and dgamma is None:
        raise RuntimeError("norm weight has no arena-backed .grad")
    if FUSE_GN_COEF and skip is None and nslots == 1 and G <= 256:
        # the coefficient step runs inside the apply pass (jg_gn_bwd_apply_fc): one launch fewer per GroupNorm backward
        if out is None:
            out = torch.empty((B, H, W, C), device=dev, dtype=x.dtype)
        adds = list(adds)
        if pooled is not None:
            if len(adds) > 1:
                raise RuntimeError("at most one full-resolution addend next to the pooled one")
            a1, s1 = pooled[1] if pooled[1] is not None else (None, 0.0)
            a2, s2 = adds[0] if adds else (None, 0.0)
        else:
            if len(adds) > 2:
                raise RuntimeError("at most two fused gradient addends")
            a1, s1 = adds[0] if len(adds) > 0 else (None, 0.0)
            a2, s2 = adds[1] if len(adds) > 1 else (None, 0.0)
        check(L.jg_gn_bwd_apply_fc(dt, int(pooled is not None), x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy),
                                   pooled[0] if pooled is not None else 1.0, ab.data_ptr(), red.data_ptr(), _p(gamma), _p(beta), _p(film),
                                   film.stride(0) if film is not None else 0, mr.data_ptr(), _p(dgamma), _p(dbeta), _p(dfilm),
                                   dfilm.stride(0) if dfilm is not None else 0, G, out.data_ptr(), _ld(out), _p(a1),
                                   _ld(a1) if a1 is not None else 0, s1, _p(a2), _ld(a2) if a2 is not None else 0, s2, B, H, W, C, act, _st()),
              "jg_gn_bwd_apply_fc")
        return out
    pqr = torch.empty((B, C, 3), device=dev, dtype=torch.float32)
    check(L.jg_gn_bwd_coef_slots(red.data_ptr(), nslots, _p(gamma), _p(beta), _p(film),
                                 film.stride(0) if film is not None else 0, mr.data_ptr(), pqr.data_ptr(), _p(dgamma),
                                 _p(dbeta), _p(dfilm), dfilm.stride(0) if dfilm is not None else 0, B, HW, C, G, _st()),
          "jg_gn_bwd_coef_slots")
    if out is None:
        out = torch.empty((B, H, W, C), device=dev, dtype=x.dtype)
    adds = list(adds)
    if pooled is not None:
        if len(adds) > 1:
            raise RuntimeError("at most one full-resolution addend next to the pooled one")
        a1, s1 = pooled[1] if pooled[1] is not None else (None, 0.0)
        a2, s2 = adds[0] if adds else (None, 0.0)
        check(L.jg_gn_bwd_apply_up(dt, x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), pooled[0], ab.data_ptr(), pqr.data_ptr(),
                                   out.data_ptr(), _ld(out), _p(a1), _ld(a1) if a1 is not None else 0, s1, _p(a2),
                                   _ld(a2) if a2 is not None else 0, s2, B, H, W, C, act, _st()), "jg_gn_bwd_apply_up")
        return out
    if len(adds) > 2:
        raise RuntimeError("at most two fused gradient addends")
    a1, s1 = adds[0] if len(adds) > 0 else (None, 0.0)
    a2, s2 = adds[1] if len(adds) > 1 else (None, 0.0)
    if skip is not None:
        dO, m, alpha = skip
        if FUSE_SKIP_BWD and m.R == 1 and m.S == 1 and m.stride == 1 and m.pad == 0:
            ok = conv_nt(dO, m.w16T, out, B=B, H=H, W=W, Cin=m.Cout, Cout=C, R=1, S=1, pad=0, stride=1, Ho=H, Wo=W, ldx=_ld(dO), ldw=m.Cout,
                         ldy=_ld(out), alpha=alpha, gn_bwd_apply=(x, dy, ab, pqr, a1, s1, a2, s2, act))
            if ok is not False:
                return out
    check(L.jg_gn_bwd_apply_ld(dt, x.data_ptr(), _ld(x), dy.data_ptr(), _ld(dy), ab.data_ptr(), pqr.data_ptr(), out.data_ptr(),
                               _ld(out), _p(a1), _ld(a1) if a1 is not None else 0, s1, _p(a2),
                               _ld(a2) if a2 is not None else 0, s2, B, HW, C, act, _st()), "jg_gn_bwd_apply_ld")
    if skip is not None:      # two-launch form: the GroupNorm-backward result enters the input gradient as its epilogue residual
        dO, m, alpha = skip
        return conv_dgrad(dO, m, x.shape, res=out, alpha=alpha)
    return out


def gn_apply_pool(x, ab, act, scale):
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_gn_apply_pool(_dt(x), x.data_ptr(), _ld(x), ab.data_ptr(), y.data_ptr(), C, B, H, W, C, act, scale, _st()),
          "jg_gn_apply_pool")
    return y


def pool2(x, scale):
    B, H, W, C = x.shape
    y = torch.empty((B, H // 2, W // 2, C), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_pool2x2_ld(_dt(x), x.data_ptr(), _ld(x), y.data_ptr(), C, B, H, W, C, scale, _st()), "jg_pool2x2_ld")
    return y


def up2(x, scale):
    B, H, W, C = x.shape
    y = torch.empty((B, H * 2, W * 2, C), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_upsample2x_ld(_dt(x), x.data_ptr(), _ld(x), y.data_ptr(), C, B, H, W, C, scale, _st()),
          "jg_upsample2x_ld")
    return y


# ---------------------------------------------------------------------------------------------------
# the schedule
# ---------------------------------------------------------------------------------------------------
class UNetExecutor:
    def __init__(self, unet):
        from .unet_generator_attn import AttentionBlock, ResBlock

        self.u = unet
        self.ResBlock, self.AttentionBlock = ResBlock, AttentionBlock
        n_in = len(unet.input_blocks)
        chans = []
        for j, blk in enumerate(unet.input_blocks):
            last = list(blk)[-1]
            if j == 0:
                chans.append(last.out_channels)
            else:
                chans.append(last.out_channel if isinstance(last, ResBlock) else last.channels)
        self.n_in = n_in
        self.cat_ch = []   # j -> (Ctot, Ca, Cb): output_blocks[n_in-1-j] consumes cat(h[Ca], hs[j][Cb])
        for j in range(n_in):
            ctot = list(unet.output_blocks[n_in - 1 - j])[0].channels
            self.cat_ch.append((ctot, ctot - chans[j], chans[j]))
        self.tape = None
        self._pool_need = {}
        self._side = None          # HIP stream of the weight-gradient launches (created with the first backward)
        self.handle = None         # torch.ops handle (fused_unet)
        self._g1x1 = []            # collected 1x1 weight gradients (GROUP_1X1_WGRAD)

    def arena_g(self):
        return self.u._jg_arena_ref.g

    # ---- weight gradients off the critical chain -----------------------------------------------------
    def wgrad(self, dy, x, m, **kw):
        """conv_wgrad on the side stream: it starts once everything launched so far on the compute stream (the producers of dy and
        x) is done.  Both operands are marked as in use by the side stream, so the caching allocator does not hand their memory to a
        later allocation of the compute stream while the kernel may still be reading it."""
        if GROUP_1X1_WGRAD and m.R == 1 and m.S == 1 and not kw.get("x_up"):
            # experiment (VERDICT r5 next #1c, JG_GROUP_1X1_WGRAD=1): the 1x1 weight gradients of the backward are collected and leave as grouped
            # launches (jg_conv2d_wgrad_tn_group) every GROUP_1X1_EVERY problems and at the end of the backward
            self._g1x1.append((dy, x, m, kw))
            if len(self._g1x1) >= GROUP_1X1_EVERY:
                self.flush_1x1()
            return
        if not WGRAD_STREAM:
            return conv_wgrad(dy, x, m, **kw)
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = _side_stream(dy.device)
            # a gradient chunk's all-reduce is ordered behind the CURRENT stream only: the weight gradients of the side stream are
            # brought in right before a chunk leaves (parallel.EarlyExchange._launch), not after every layer
            import weakref

            parallel.PRE_LAUNCH_HOOKS.append(weakref.WeakMethod(self.wgrad_join))      # weak: the hooks die with this executor
            # ... and an EARLY chunk (launched from inside this backward) is enqueued FROM the side stream: ordered behind every weight
            # gradient launched so far without the compute stream waiting for them (parallel.LAUNCH_CONTEXTS)
            parallel.LAUNCH_CONTEXTS.append(weakref.WeakMethod(self.exchange_context))
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            conv_wgrad(dy, x, m, **kw)
        dy.record_stream(self._side)
        x.record_stream(self._side)

    def flush_1x1(self):
        if not self._g1x1:
            return
        items, self._g1x1 = self._g1x1, []
        main = torch.cuda.current_stream()
        side = self._side if (WGRAD_STREAM and self._side is not None) else main
        if side is not main:
            side.wait_stream(main)
        with torch.cuda.stream(side), ops.deferred_wgrads():
            for dy, x, m, kw in items:
                conv_wgrad(dy, x, m, **kw)
        for dy, x, m, kw in items:
            if side is not main:
                dy.record_stream(side)
                x.record_stream(side)

    @contextlib.contextmanager
    def exchange_context(self):
        """launch context of a gradient chunk that leaves from inside the backward: the side stream first waits for the compute stream
        (norm / bias gradients of the chunk are produced there), then the collective is enqueued with the side stream current"""
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            yield

    def wgrad_join(self):
        """the compute stream waits for every weight gradient launched so far (end of the backward; before an all-reduce chunk)"""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, xin, emb):
        u = self.u
        B, H, W, _ = xin.shape
        dev = xin.device
        self.emb = emb
        key = (B, H, W)
        self.pool = _Pool(self._pool_need.get(key, B * NSLOT * 2 * 40000), dev)
        self.tape = []
        self.cats = {}
        tape = self.tape

        def cat_slot(j, half):
            def make(Bq, Hq, Wq, Cq):
                ctot, ca, cb = self.cat_ch[j]
                if j not in self.cats:
                    self.cats[j] = (torch.empty((Bq, Hq, Wq, ctot), device=dev, dtype=xin.dtype), self.pool.take(Bq, ctot))
                buf, st = self.cats[j]
                assert buf.shape[1] == Hq and buf.shape[2] == Wq, "concat halves disagree on the spatial size"
                lo, hi = (0, ca) if half == "a" else (ca, ctot)
                assert hi - lo == Cq, (j, half, Cq, lo, hi)
                return buf[..., lo:hi], st[:, :, lo:hi, :]
            return make

        def fresh(Bq, Hq, Wq, Cq):
            return torch.empty((Bq, Hq, Wq, Cq), device=dev, dtype=xin.dtype), self.pool.take(Bq, Cq)

        # stem
        stem = list(u.input_blocks[0])[0].meta
        t, st = cat_slot(0, "b")(B, H, W, stem.Cout)
        conv_fwd(xin, stem, out=t, stats=st)
        tape.append(dict(kind="stem", xin=xin, m=stem, add_hs=None, cat_j=None, in_id=None))
        h = Act(t, st, H * W, 0, hs_j=0)

        def run_layers(layers, h, last_dest, hs_j):
            for li, layer in enumerate(layers):
                is_last = li == len(layers) - 1
                dest = last_dest if is_last else fresh
                if isinstance(layer, self.ResBlock):
                    h = self.res_fwd(layer, h, dest)
                elif isinstance(layer, self.AttentionBlock):
                    h = self.attn_fwd(layer, h, dest)
                else:
                    raise NotImplementedError(type(layer))
                if is_last:
                    h.hs_j = hs_j
            return h

        for j in range(1, self.n_in):
            h = run_layers(list(u.input_blocks[j]), h, cat_slot(j, "b"), j)
        h = run_layers(list(u.middle_block), h, cat_slot(self.n_in - 1, "a"), None)
        n_out = len(u.output_blocks)
        for k, blk in enumerate(u.output_blocks):
            j = self.n_in - 1 - k
            buf, st = self.cats[j]
            X = Act(buf, st, buf.shape[1] * buf.shape[2], ("cat", j, h.pid))
            h = run_layers(list(blk), X, cat_slot(j - 1, "a") if k < n_out - 1 else fresh, None)
        # head
        gn, head = u.out[0].norm, u.out[2].meta
        ab, mr = gn_coef(h.st, h.hw, gn.weight, gn.bias, None, gn.num_groups, gn.eps)
        hn = gn_apply(h.t, ab, JG_ACT_SILU)
        out = conv_fwd(hn, head)
        tape.append(dict(kind="head", x=h.t, ab=ab, mr=mr, hn=hn, gn=gn, m=head, in_id=h.pid, add_hs=h.hs_j, cat_j=None))
        self._pool_need[key] = self.pool.off   # exact size from the second forward of a shape on
        self.cats = None
        self.pool = None
        # the tape belongs to THIS call (several forwards may be alive before their backwards, e.g. the
        # student / no-grad teacher pair of the consistency model)
        state = (self.tape, self.emb, key)
        self.tape = self.emb = None
        return out, state

    def _in_fields(self, X):
        if isinstance(X.pid, tuple):
            _, j, a_id = X.pid
            return dict(in_id=None, add_hs=None, cat_j=j, cat_a_id=a_id, Ca=self.cat_ch[j][1])
        return dict(in_id=X.pid, add_hs=X.hs_j, cat_j=None)

    def res_fwd(self, rb, X, dest):
        if rb.dropout and rb.training:
            raise NotImplementedError("dropout > 0 inside ResBlock is not implemented")
        x = X.t
        B, H, W, Cin = x.shape
        Cout = rb.out_channel
        gn1, c1m = rb.in_layers[0].norm, rb.in_layers[2].meta
        gn2, c2m = rb.out_layers[0].norm, rb.out_layers[3].meta
        ab1, mr1 = gn_coef(X.st, X.hw, gn1.weight, gn1.bias, None, gn1.num_groups, gn1.eps)
        fuse_down = rb.updown and rb.down and FUSE_DOWN_POOL
        sk_pre = None
        if FUSE_SKIP_APPLY and not rb.updown and not isinstance(rb.skip_connection, nn.Identity):
            both = conv1x1_gn_apply(x, rb.skip_connection.meta, ab1, JG_ACT_SILU)      # x is read once for its two consumers
            if both is not None:
                h1, sk_pre = both
        if sk_pre is None:
            h1 = None if fuse_down else gn_apply(x, ab1, JG_ACT_SILU)
        st1 = self.pool.take(B, Cout)
        Ho, Wo = H, W
        if rb.updown:
            Ho, Wo = (H * 2, W * 2) if rb.up else (H // 2, W // 2)
            if rb.up and rb.efficient:     # conv before the upsample (reference :239-242)
                a1 = h1
                c1 = conv_fwd(h1, c1m, stats=st1)
                if not NORM_BEFORE_UP:
                    c1 = up2(c1, 1.0)
                hw1 = H * W                # statistics of the source: same mean / variance
            else:
                if fuse_down:
                    a1 = gn_apply_pool(x, ab1, JG_ACT_SILU, 0.25)     # pool(act(norm(x))) in one pass
                else:
                    a1 = up2(h1, 1.0) if rb.up else pool2(h1, 0.25)
                c1 = conv_fwd(a1, c1m, stats=st1)
                hw1 = Ho * Wo
            identity = isinstance(rb.skip_connection, nn.Identity)
            # identity skip of an up-block: the convolution epilogue reads x through the upsample map (res_mode 1)
            res_up = rb.up and identity and RES_UP_ON_READ
            xs = x if res_up else (up2(x, 1.0) if rb.up else pool2(x, 0.25))
        else:
            res_up = False
            a1, xs, hw1 = h1, x, H * W
            c1 = conv_fwd(h1, c1m, stats=st1)
        off, n = rb.emb_slice
        film = self.emb[:, off:off + n]
        ab2, mr2 = gn_coef(st1, hw1, gn2.weight, gn2.bias, film, gn2.num_groups, gn2.eps)
        h2 = gn_apply(c1, ab2, JG_ACT_SILU)
        low2 = rb.updown and rb.up and rb.efficient and NORM_BEFORE_UP
        if low2:
            # GroupNorm + FiLM + SiLU are point-wise given the statistics, and a nearest-neighbour upsample leaves the statistics
            # unchanged: normalise the LOW-resolution conv output and upsample the activated tensor (a quarter of the pass; the
            # backward pools the gradient first and runs the GroupNorm backward at low resolution as well -- same mathematics)
            # ... and the upsampled tensor itself is never written when conv2 / its weight gradient can read the low-resolution one
            # through the upsample index map (x_mode 1 of the halo-resident kernels)
            h2_up = X_UP_ON_READ and halo_ok(c2m, Ho, Wo)
            if not h2_up:
                h2 = up2(h2, 1.0)
        else:
            h2_up = False
        skipw = 1.0 / math.sqrt(2) if rb.efficient else 1.0
        identity = isinstance(rb.skip_connection, nn.Identity)
        sk = xs if identity else (sk_pre if sk_pre is not None else conv_fwd(xs, rb.skip_connection.meta))
        out_t, out_st = dest(B, Ho, Wo, Cout)
        wfold = subpixel_fold(c2m, h2.dtype) if (h2_up and SUBPIXEL_CONV and subpixel_ok(c2m, Ho, Wo)) else None
        conv_fwd(h2, c2m, out=out_t, res=sk, res_scale=skipw, stats=out_st, res_up=res_up, x_up=h2_up, wfold=wfold)
        rec = dict(kind="res", rb=rb, x=x, ab1=ab1, mr1=mr1, a1=a1, c1=c1, ab2=ab2, mr2=mr2, h2=h2, film=film,
                   xs=None if identity else xs, skipw=skipw, identity=identity, low2=low2, h2_up=h2_up)
        rec.update(self._in_fields(X))
        self.tape.append(rec)
        return Act(out_t, out_st, Ho * Wo, len(self.tape) - 1)

    def attn_fwd(self, blk, X, dest):
        x = X.t
        B, H, W, C = x.shape
        T = H * W
        ab, mr = gn_coef(X.st, X.hw, None, None, None, C, 1e-5)      # InstanceNorm1d over T, no affine
        xn = gn_apply(x, ab, JG_ACT_NONE)
        qkv = conv_fwd(xn.view(B, 1, T, C), blk.qkv.meta)
        a, P = attn_core_fwd(qkv.view(B, T, 3 * C), blk.num_heads)
        out_t, out_st = dest(B, H, W, C)
        conv_fwd(a.view(B, 1, T, C), blk.proj_out.meta, out=out_t.view(B, 1, T, C), res=x.view(B, 1, T, C), res_scale=1.0,
                 stats=out_st)
        rec = dict(kind="attn", blk=blk, x=x, ab=ab, mr=mr, xn=xn, qkv=qkv, P=P, a=a)
        rec.update(self._in_fields(X))
        self.tape.append(rec)
        return Act(out_t, out_st, T, len(self.tape) - 1)

    # ---- backward ----------------------------------------------------------------------------------
    def backward(self, state, dout, need_dx=False):
        tape, self.emb, self._key = state
        self.dxin = None
        if tape is None:
            raise RuntimeError("UNetExecutor.backward: this forward's tape was already consumed")
        self.demb = torch.empty_like(self.emb)
        bkey = ("bwd",) + self._key
        self.bpool = _Pool(self._pool_need.get(bkey, 2 * self._pool_need[self._key] + 65536), self.demb.device)
        dacts, dhs = {}, {}
        for idx in range(len(tape) - 1, -1, -1):
            rec = tape[idx]
            tape[idx] = None
            dO = dout if rec["kind"] == "head" else dacts.pop(idx)
            if rec["kind"] == "stem":
                self.wgrad(dO, rec["xin"], rec["m"])
                if need_dx:
                    self.dxin = conv_dgrad(dO, rec["m"], rec["xin"].shape)
                self._grads_final(rec)
                continue
            adds = []
            if rec.get("add_hs") is not None:
                adds.append((dhs.pop(rec["add_hs"]), 1.0))
            if rec["kind"] == "head":
                dX = self.head_bwd(rec, dO, adds)
            elif rec["kind"] == "res":
                dX = self.res_bwd(rec, dO, adds)
            else:
                dX = self.attn_bwd(rec, dO, adds)
            self._grads_final(rec)     # data parallel: chunks of the gradient arena start their all-reduce here
            if rec["cat_j"] is not None:
                ca = rec["Ca"]
                dacts[rec["cat_a_id"]] = dX[..., :ca]
                dhs[rec["cat_j"]] = dX[..., ca:]
            else:
                dacts[rec["in_id"]] = dX
        assert not dacts and not dhs, (list(dacts), list(dhs))
        self.flush_1x1()
        self.wgrad_join()
        self._pool_need[bkey] = max(self.bpool.off, 64)
        self.bpool = None
        demb, self.demb, self.emb = self.demb, None, None
        return demb

    def _grads_final(self, rec):
        parallel.grads_final(_own_params(rec))

    def head_bwd(self, rec, dO, adds):
        gn, m = rec["gn"], rec["m"]
        dhn, red = conv_dgrad(dO, m, rec["hn"].shape, gn=(rec["x"], rec["ab"], JG_ACT_SILU), pool=self.bpool)
        self.wgrad(dO, rec["hn"], m)
        return gn_bwd(rec["x"], dhn, rec["ab"], rec["mr"], gn.weight, gn.bias, None, gn.num_groups, JG_ACT_SILU, adds=adds,
                      red=red, pool=self.bpool)

    def res_bwd(self, rec, dO, adds):
        rb = rec["rb"]
        x = rec["x"]
        gn1, c1m = rb.in_layers[0].norm, rb.in_layers[2].meta
        gn2, c2m = rb.out_layers[0].norm, rb.out_layers[3].meta
        skipw = rec["skipw"]
        # conv2
        if WGRAD_FIRST:
            self.wgrad(dO, rec["h2"], c2m, x_up=rec["h2_up"])
        if rec.get("low2"):
            hb, hh, hw_, hc = rec["h2"].shape
            full = (hb, 2 * hh, 2 * hw_, hc) if rec["h2_up"] else (hb, hh, hw_, hc)
            # adjoint of the upsample behind GroupNorm 2: pooled in the input-gradient epilogue when the halo kernel runs the layer
            if POOL_IN_DGRAD and halo_ok(c2m, full[1], full[2]):
                dh2, red2 = conv_dgrad(dO, c2m, full, pool_out=True), None
            else:
                dh2, red2 = pool2(conv_dgrad(dO, c2m, full), 1.0), None
        else:
            dh2, red2 = conv_dgrad(dO, c2m, rec["h2"].shape, gn=(rec["c1"], rec["ab2"], JG_ACT_SILU), pool=self.bpool)
        if not WGRAD_FIRST:
            self.wgrad(dO, rec["h2"], c2m, x_up=rec["h2_up"])
        # GroupNorm 2 (+FiLM +SiLU)
        off, n = rb.emb_slice
        dc1 = gn_bwd(rec["c1"], dh2, rec["ab2"], rec["mr2"], gn2.weight, gn2.bias, rec["film"], gn2.num_groups, JG_ACT_SILU,
                     dfilm=self.demb[:, off:off + n], red=red2, pool=self.bpool)
        del dh2
        if rb.up and rb.efficient and not rec.get("low2"):
            dc1 = pool2(dc1, 1.0)          # backward of the nearest upsample that follows conv1
        # conv1
        direct = (not rb.updown) or (rb.up and rb.efficient)   # da1 IS the output-gradient of GroupNorm 1
        if WGRAD_FIRST:
            self.wgrad(dc1, rec["a1"], c1m)
        if direct:
            da1, red1 = conv_dgrad(dc1, c1m, rec["a1"].shape, gn=(x, rec["ab1"], JG_ACT_SILU), pool=self.bpool)
        else:
            da1, red1 = conv_dgrad(dc1, c1m, rec["a1"].shape), None
        if not WGRAD_FIRST:
            self.wgrad(dc1, rec["a1"], c1m)
        del dc1
        pooled = None
        if rb.down and FUSE_DOWN_POOL:
            dh1 = da1                      # stays at the pooled resolution; gn_bwd applies the pool's adjoint while reading
            pooled = (0.25, (dO, skipw * 0.25) if rec["identity"] else None)
        elif rb.down:
            dh1 = up2(da1, 0.25)
        elif rb.up and not rb.efficient:
            dh1 = pool2(da1, 1.0)
        else:
            dh1 = da1
        # skip path + GroupNorm 1
        adds = list(adds)
        if rec["identity"]:
            if not rb.updown:
                adds.append((dO, skipw))
            elif rb.down:
                if pooled is None:
                    adds.append((up2(dO, skipw * 0.25), 1.0))
            else:
                adds.append((pool2(dO, skipw), 1.0))
            return gn_bwd(x, dh1, rec["ab1"], rec["mr1"], gn1.weight, gn1.bias, None, gn1.num_groups, JG_ACT_SILU, adds=adds,
                          red=red1, pooled=pooled, pool=self.bpool)
        if rb.updown:
            raise NotImplementedError("resampling ResBlock with a 1x1 skip convolution")
        skm = rb.skip_connection.meta
        self.wgrad(dO, rec["xs"], skm, alpha=skipw, dbias_scale=skipw)
        # skipw * (dO . Wskip) + GroupNorm-backward(x, dh1) + addends: one launch behind the reduction + coefficient steps
        return gn_bwd(x, dh1, rec["ab1"], rec["mr1"], gn1.weight, gn1.bias, None, gn1.num_groups, JG_ACT_SILU, adds=adds,
                      red=red1, pool=self.bpool, skip=(dO, skm, skipw))

    def attn_bwd(self, rec, dO, adds):
        blk = rec["blk"]
        x = rec["x"]
        B, H, W, C = x.shape
        T = H * W
        dO4 = dO.view(B, 1, T, C) if dO.dim() == 4 else dO
        a4 = rec["a"].view(B, 1, T, C)
        da = conv_dgrad(dO4, blk.proj_out.meta, a4.shape)
        self.wgrad(dO4, a4, blk.proj_out.meta)
        dqkv = attn_core_bwd(rec["qkv"].view(B, T, 3 * C), rec["P"], da.view(B, T, C), blk.num_heads, rec["a"])
        xn4 = rec["xn"].view(B, 1, T, C)
        dxn, red = conv_dgrad(dqkv.view(B, 1, T, 3 * C), blk.qkv.meta, xn4.shape, gn=(x.view(B, 1, T, C), rec["ab"], JG_ACT_NONE),
                              pool=self.bpool)
        self.wgrad(dqkv.view(B, 1, T, 3 * C), xn4, blk.qkv.meta)
        adds = list(adds) + [(dO, 1.0)]
        return gn_bwd(x, dxn.view(B, H, W, C), rec["ab"], rec["mr"], None, None, None, C, JG_ACT_NONE, adds=adds, red=red, pool=self.bpool)


def _own_params(rec):
    """parameters whose gradient is complete once this record's backward has been launched (the embedding projections of the
    ResBlocks are NOT: their gradient comes from the stacked linear behind the node)"""
    kind = rec["kind"]
    if kind == "stem":
        objs = [rec["m"]]
    elif kind == "head":
        objs = [rec["gn"], rec["m"]]
    elif kind == "res":
        rb = rec["rb"]
        objs = [rb.in_layers[0].norm, rb.in_layers[2].meta, rb.out_layers[0].norm, rb.out_layers[3].meta]
        if not rec["identity"]:
            objs.append(rb.skip_connection.meta)
    else:
        objs = [rec["blk"].qkv.meta, rec["blk"].proj_out.meta]
    out = []
    for o in objs:
        for nm in ("weight", "bias"):
            prm = getattr(o, nm, None)
            if prm is not None:
                out.append(prm)
    return out


class _FusedUNetFn(JGFunction):
    """(xin, emb_all) -> UNet output; `exe` carries the tape between forward and backward."""

    @staticmethod
    def forward(ctx, xin, emb, exe):
        ctx.exe = exe
        out, state = exe.forward(xin, emb)
        ctx.state = state if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None   # inference keeps nothing
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        state, ctx.state = ctx.state, None
        demb = ctx.exe.backward(state, dout.contiguous(), ctx.needs_input_grad[0])
        dx, ctx.exe.dxin = ctx.exe.dxin, None
        return dx, demb, None


# ---- the fused node as a pair of torch.ops (round 6; VERDICT r5 missing #3 / next #6) ---------------------------------------------------------
# north_star: the kernels are "exposed to the Python host as torch.ops"; precedent in the reference: models/modules/op/upfirdn2d.py:19-167 (a
# custom op whose backward is a second custom op).  `jg355::unet_fused` is the forward of the whole UNet on the fused schedule (what bench.py
# times), `jg355::unet_fused_bwd` its backward; both are torch.library custom ops with fake kernels, the pair is tied by register_autograd, and
# this is what `UNet.forward` -- i.e. `optimize_parameters()` -- calls by default (JG_FUSED_TORCH_OPS=0: the autograd.Function of rounds 1-5,
# same executor behind it).
#   * the executor (module tree, launch schedule) is addressed by an integer HANDLE; the weights it reads are passed as the arena's 16-bit
#     working copies (`w16`, `w16T`: the data dependence is visible to the dispatcher), the fp32 gradient arena it accumulates into is an
#     argument of the backward op declared in `mutates_args`;
#   * the tape (saved activations, statistics, the pool of GroupNorm sums) stays with the executor under an integer id that the forward
#     returns as a 1-element int64 tensor; the backward op consumes it, a finalizer on that tensor drops an unused tape.
FUSED_TORCH_OPS = os.environ.get("JG_FUSED_TORCH_OPS", "1") != "0"
GROUP_1X1_WGRAD = os.environ.get("JG_GROUP_1X1_WGRAD", "0") != "0"
GROUP_1X1_EVERY = int(os.environ.get("JG_GROUP_1X1_EVERY", "64"))
_EXES, _TAPES, _NEXT = {}, {}, [1]


@torch.library.custom_op("jg355::unet_fused", mutates_args=())
def _op_unet_fused(xin: torch.Tensor, emb: torch.Tensor, w16: torch.Tensor, w16T: torch.Tensor, handle: int, keep: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """UNet forward on the fused schedule: (xin [B,H,W,Cpad] 16-bit NHWC, emb = stacked FiLM projections) -> (out [B,H,W,8], tape id)"""
    exe = _EXES[handle]
    out, state = exe.forward(xin, emb)
    tid = 0
    if keep:
        tid = _NEXT[0]
        _NEXT[0] += 1
        _TAPES[tid] = state
    return out, torch.tensor([tid], dtype=torch.int64)


@_op_unet_fused.register_fake
def _(xin, emb, w16, w16T, handle, keep):
    B, H, W, _c = xin.shape
    return xin.new_empty((B, H, W, 8)), torch.empty(1, dtype=torch.int64)


@torch.library.custom_op("jg355::unet_fused_bwd", mutates_args=("grad",))
def _op_unet_fused_bwd(dout: torch.Tensor, tape: torch.Tensor, grad: torch.Tensor, handle: int, need_dx: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """backward of jg355::unet_fused: -> (d xin (empty when not needed), d emb); every parameter gradient is ACCUMULATED into `grad`, the fp32
    gradient arena of the network (weight gradients on the side stream, joined before this op returns)"""
    exe = _EXES[handle]
    if grad.data_ptr() != exe.arena_g().data_ptr():
        raise ValueError("jg355::unet_fused_bwd: `grad` must be the fp32 gradient arena of the network behind `handle` (the kernels accumulate "
                         "into the arena the executor was built on)")
    state = _TAPES.pop(int(tape[0]))
    demb = exe.backward(state, dout.contiguous(), need_dx)
    dx, exe.dxin = exe.dxin, None
    return (dx if dx is not None else dout.new_empty(0)), demb


@_op_unet_fused_bwd.register_fake
def _(dout, tape, grad, handle, need_dx):
    exe = _EXES[handle]
    B, H, W, _c = dout.shape
    return (dout.new_empty((B, H, W, exe.in_pad)) if need_dx else dout.new_empty(0)), dout.new_empty(exe.emb_shape, dtype=torch.float32)


def _fused_setup(ctx, inputs, output):
    ctx.handle = inputs[4]
    ctx.need_dx = ctx.needs_input_grad[0]
    ctx.save_for_backward(output[1])


def _fused_backward(ctx, dout, _dtape):
    (tape,) = ctx.saved_tensors
    dx, demb = torch.ops.jg355.unet_fused_bwd(dout, tape, _EXES[ctx.handle].arena_g(), ctx.handle, ctx.need_dx)
    return (dx if ctx.need_dx else None), demb, None, None, None, None


_op_unet_fused.register_autograd(_fused_backward, setup_context=_fused_setup)


def fused_unet(unet, xin, emb_all):
    exe = getattr(unet, "_jg_executor", None)
    if exe is None:
        exe = unet._jg_executor = UNetExecutor(unet)
    if FUSED_TORCH_OPS:
        if exe.handle is None:
            exe.handle = len(_EXES) + 1
            _EXES[exe.handle] = exe
        exe.in_pad, exe.emb_shape = xin.shape[-1], tuple(emb_all.shape)
        arena = unet._jg_arena_ref
        keep = torch.is_grad_enabled() and (xin.requires_grad or emb_all.requires_grad)
        out, tape = torch.ops.jg355.unet_fused(xin, emb_all, arena.w16, arena.w16T, exe.handle, keep)
        if keep:
            weakref.finalize(tape, _TAPES.pop, int(tape[0]), None)      # a forward whose graph is dropped without a backward
        return out
    return _FusedUNetFn.apply(xin, emb_all, exe)

"""Host-side operators over libjg355.so: thin launch wrappers, the autograd formulas that chain
the hand-written forward/backward kernels, and the `torch.ops.jg355.*` registrations.

Conventions
  * activations: 16-bit (torch.float16 / torch.bfloat16), logical AND physical NHWC
    `[B, H, W, C]` (or `[B, T, C]`), contiguous, C % 8 == 0;
  * parameters: fp32 views into a ParamArena (joligen_amd/arena.py); kernels ACCUMULATE parameter
    gradients straight into `param.grad` (an arena view) -- the autograd Functions return None
    for parameter inputs, which are passed only so that autograd tracks them;
  * every kernel is launched on torch's current stream; nothing here synchronises.

PyTorch is used for device memory (caching allocator), streams and the autograd tape only.
"""
from __future__ import annotations

import ctypes as C
import math
import os

from typing import Optional, Tuple

import torch

from ._autograd import JGFunction
from . import _lib
from ._lib import (JG_ACT_LRELU, JG_ACT_NONE, JG_ACT_RELU, JG_ACT_SILU, JG_ACT_TANH, JG_OUT_ATOMIC_F32, JG_OUT_STORE_T, ConvArgs,
                   WgradArgs, check)

_DT = {torch.float16: _lib.JG_F16, torch.bfloat16: _lib.JG_BF16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"joligen_amd activations must be float16/bfloat16, got {t.dtype}") from None


# torch.cuda.current_stream() builds a Stream object per call (~8 us): too slow for a per-launch lookup.  The raw accessor is a private
# symbol: a torch build without it (or without CUDA / HIP at all) falls back to the public API instead of failing at import time
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda dev: torch.cuda.current_stream(dev).cuda_stream)
_cur_device = torch.cuda.current_device


def _st() -> int:
    """raw hipStream_t of torch's current stream on the current device"""
    return _raw_stream(_cur_device())


def _p(t):
    return None if t is None else t.data_ptr()


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("joligen_amd ops run on the GPU only (no CPU fallback); got a CPU tensor")


# ======================================================================================
# raw launchers
# ======================================================================================
# bench.py sets this to a list to time every launch of the dominant kernel with HIP events
# (start/stop recorded on the launch stream); entries: (kernel, ev_start, ev_stop, algorithmic flops)
KERNEL_TIMING = None
# fused (flash-style) attention kernel for head dim 32; JG_FLASH_ATTENTION=0 selects the GEMM + softmax pipeline
FLASH_ATTENTION = __import__("os").environ.get("JG_FLASH_ATTENTION", "1") != "0"

def _halo_ok(nbatch, H, W, Cin, Cout, R, S, pad, stride):
    """mirror of the dispatch conditions of conv_halo.hip / wgrad_halo.hip (labels for bench.py only)"""
    return (nbatch == 1 and R == 3 and S == 3 and pad == 1 and stride == 1 and Cin % 64 == 0 and Cout % 64 == 0
            and H % 16 == 0 and W % 16 == 0)


def _conv_kernel_name(nbatch, H, W, Cin, Cout, R, S, pad, stride, M=0):
    if _halo_ok(nbatch, H, W, Cin, Cout, R, S, pad, stride):
        return "conv3x3_halo_kernel"
    if (nbatch == 1 and R == 1 and S == 1 and pad == 0 and stride == 1 and Cin % 32 == 0 and Cin <= 256 and Cout % 64 == 0 and M >= 65536
            and M % 16 == 0 and os.environ.get("JG_CONV1X1", "1") != "0"):
        return "conv1x1_stream_kernel"
    if (nbatch == 1 and R == 3 and S == 3 and pad == 1 and stride == 1 and Cin == 8 and Cout % 64 == 0 and M >= 65536 and M % 16 == 0
            and os.environ.get("JG_CONV1X1", "1") != "0"):
        return "conv3x3_c8_stream_kernel"
    return "conv_nt_glds_kernel<256,64,64,4,1>" if Cout <= 64 else "conv_nt_glds_kernel<128,128,64,2,2>"


def _wgrad_kernel_name(nbatch, H, W, Cin, Cout, R, S, pad, stride, out_mode):
    if out_mode == JG_OUT_ATOMIC_F32 and _halo_ok(nbatch, H, W, Cin, Cout, R, S, pad, stride):
        return "wgrad3x3_halo_kernel"
    return "wgrad_tn_tr_kernel<1>" if Cout <= 64 else "wgrad_tn_tr_kernel<2>"


_CONV_WS = {}               # (device index, stream) -> fp32 scratch of the split-K form of the generic conv kernel (jg_conv_args.ws)
CONV_WS_BYTES = 32 << 20


def _conv_ws(dev):
    key = (dev.index, _raw_stream(dev.index))
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.empty(CONV_WS_BYTES // 4, device=dev, dtype=torch.float32)
    return ws


def conv_nt(x, w, y, *, B, H, W, Cin, Cout, R, S, pad, stride, Ho, Wo, ldx, ldw, ldy, bias=None, res=None,
            ldres=0, alpha=1.0, res_scale=1.0, out_f32=False, nbatch=1, nh=1, sx=(0, 0), sw=(0, 0), sy=(0, 0),
            sr=(0, 0), x_off=0, w_off=0, y_off=0, dtype=None, stats=None, ldstats=0, stats_slots=1, gn_reduce=None, pad_mode=0, res_mode=0, x_mode=0, y_mode=0,
            apply=None, gn_bwd_apply=None):
    """jg_conv2d_nt with element offsets into the operand tensors.
    apply = (ab, y_norm, ldyn, act): jg_conv1x1_gn_apply -- the launch also writes act(a x + b) of its input; returns False (nothing
    launched) when the shape is not the streaming 1x1 kernel's.
    gn_bwd_apply = (gx, gdy, ab, pqr, add1, scale1, add2, scale2, act): jg_conv1x1_gn_bwd_apply -- the epilogue adds the GroupNorm-backward
    apply step of the output tensor; returns False likewise."""
    a = ConvArgs()
    es = 2
    a.x = x.data_ptr() + x_off * es
    a.w = w.data_ptr() + w_off * es
    a.y = y.data_ptr() + y_off * (4 if out_f32 else es)
    a.bias = _p(bias)
    a.res = _p(res)
    a.B, a.H, a.W, a.Cin, a.Cout, a.R, a.S, a.pad, a.stride, a.Ho, a.Wo = B, H, W, Cin, Cout, R, S, pad, stride, Ho, Wo
    a.ldx, a.ldw, a.ldy, a.ldres = ldx, ldw, ldy, ldres
    a.nbatch, a.nh = nbatch, nh
    a.sxb, a.sxh = sx
    a.swb, a.swh = sw
    a.syb, a.syh = sy
    a.srb, a.srh = sr
    a.alpha, a.res_scale, a.out_f32 = alpha, res_scale, int(out_f32)
    a.stats, a.ldstats, a.stats_slots = _p(stats), ldstats, stats_slots
    a.pad_mode = pad_mode
    a.res_mode = res_mode
    a.x_mode = x_mode
    a.y_mode = y_mode
    if B * Ho * Wo <= 16384 and R * S * Cin >= 1024:      # few output tiles, long reduction: let the library cut the K loop (needs scratch)
        ws = _conv_ws(x.device)
        a.ws, a.ws_bytes = ws.data_ptr(), CONV_WS_BYTES
    if gn_reduce is not None:   # (norm input x, pixel stride, ab coefficients, act): GroupNorm-backward reductions in the epilogue
        gx, gldx, gab, gact = gn_reduce
        a.stats_mode, a.gn_x, a.gn_ldx, a.gn_ab, a.gn_act = 1, gx.data_ptr(), gldx, gab.data_ptr(), gact
    if KERNEL_TIMING is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()  # current stream == the stream the kernel is launched on
    if gn_bwd_apply is not None:
        gx, gdy, gab, gpqr, a1, s1, a2, s2, gact = gn_bwd_apply
        rc = _lib.lib().jg_conv1x1_gn_bwd_apply(dtype if dtype is not None else _dt(x), C.byref(a), gx.data_ptr(), gx.stride(-2), gdy.data_ptr(),
                                                gdy.stride(-2), gab.data_ptr(), gpqr.data_ptr(), _p(a1), a1.stride(-2) if a1 is not None else 0, s1,
                                                _p(a2), a2.stride(-2) if a2 is not None else 0, s2, gact, _st())
        if rc == _lib.JG_ERR_UNSUPPORTED:
            return False
        check(rc, "jg_conv1x1_gn_bwd_apply")
    elif apply is not None:
        ab, yn, ldyn, act = apply
        rc = _lib.lib().jg_conv1x1_gn_apply(dtype if dtype is not None else _dt(x), C.byref(a), ab.data_ptr(), yn.data_ptr(), ldyn, act, _st())
        if rc == _lib.JG_ERR_UNSUPPORTED:
            return False
        check(rc, "jg_conv1x1_gn_apply")
    else:
        check(_lib.lib().jg_conv2d_nt(dtype if dtype is not None else _dt(x), C.byref(a), _st()), "jg_conv2d_nt")
    if KERNEL_TIMING is not None:
        ev1.record()
        ran = _lib.lib().jg_last_kernel().decode()      # the instance the dispatch picked ("" where the site records none)
        if x_mode == 2:    # sub-pixel form: 4 of the 9 tap-MACs per output pixel are executed -- count the MFMA work actually done
            KERNEL_TIMING.append((ran or "conv3x3_halo_kernel<subpixel>", ev0, ev1, 2.0 * B * Ho * Wo * Cout * 4 * Cin, (nbatch, B, Ho, Wo, Cin, Cout, 2),
                                  2.0 * nbatch * (B * (Ho // 2) * (Wo // 2) * Cin + B * Ho * Wo * Cout) + 2.0 * 16 * Cin * Cout))
        else:
            KERNEL_TIMING.append((ran or _conv_kernel_name(nbatch, H, W, Cin, Cout, R, S, pad, stride, B * Ho * Wo if (stats is None or (R == 3 and Cin == 8 and gn_reduce is None)) else 0), ev0, ev1,
                                  2.0 * nbatch * B * Ho * Wo * Cout * R * S * Cin, (nbatch, B, Ho, Wo, Cin, Cout, R),
                                  # algorithmic HBM bytes: every input / output / residual element once (16-bit), the weights once; the fused
                                  # launches also move the normalised copy of the input (apply) or gx, gdy and the addends (gn_bwd_apply)
                                  2.0 * nbatch * (B * H * W * Cin + B * Ho * Wo * Cout * (2 if res is not None else 1)) + 2.0 * R * S * Cin * Cout
                                  + (2.0 * B * H * W * Cin if apply is not None else 0.0)
                                  + (2.0 * B * Ho * Wo * Cout * (2 + (gn_bwd_apply[4] is not None) + (gn_bwd_apply[6] is not None)) if gn_bwd_apply is not None else 0.0)))


def wgrad_tn(dy, x, dw, *, B, H, W, Cin, Cout, R, S, pad, stride, Ho, Wo, lddy, ldx, lddw, dbias=None, Cin_out=0,
             Cout_out=0, splitk=1, nbatch=1, nh=1, sdy=(0, 0), sx=(0, 0), sdw=(0, 0), alpha=1.0,
             out_mode=JG_OUT_ATOMIC_F32, dy_off=0, x_off=0, dw_off=0, dbias_scale=0.0, pad_mode=0, x_mode=0, defer=True):
    a = WgradArgs()
    a.dy = dy.data_ptr() + dy_off * 2
    a.x = x.data_ptr() + x_off * 2
    a.dw = dw.data_ptr() + dw_off * dw.element_size()
    a.dbias = _p(dbias)
    a.B, a.H, a.W, a.Cin, a.Cout, a.R, a.S, a.pad, a.stride, a.Ho, a.Wo = B, H, W, Cin, Cout, R, S, pad, stride, Ho, Wo
    a.Cin_out, a.Cout_out = Cin_out, Cout_out
    a.lddy, a.ldx, a.lddw = lddy, ldx, lddw
    a.nbatch, a.nh, a.splitk = nbatch, nh, splitk
    a.sdyb, a.sdyh = sdy
    a.sxb, a.sxh = sx
    a.sdwb, a.sdwh = sdw
    a.alpha, a.out_mode, a.dbias_scale = alpha, out_mode, dbias_scale
    a.pad_mode = pad_mode
    a.x_mode = x_mode
    # never under the torch.ops boundary: jg355::conv2d_wgrad hands a TEMPORARY dw / dbias back to autograd, which accumulates it into .grad at
    # once -- a launch deferred to the flush would write into a tensor nobody reads any more (silently zero gradients)
    if (defer and WGRAD_DEFER is not None and nbatch == 1 and out_mode == JG_OUT_ATOMIC_F32 and pad_mode == 0 and x_mode == 0 and KERNEL_TIMING is None
            and not TORCH_OPS_BOUNDARY and not _IN_OP):
        WGRAD_DEFER.append((_dt(dy), a, (dy, x, dw, dbias)))       # launched with its peers by flush_deferred_wgrads(); operands kept alive
        if _WGRAD_SIDE is not None and len(WGRAD_DEFER) >= _WGRAD_SIDE[1]:
            flush_deferred_wgrads()
        return
    if KERNEL_TIMING is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(_lib.lib().jg_conv2d_wgrad_tn(_dt(dy), C.byref(a), _st()), "jg_conv2d_wgrad_tn")
    if KERNEL_TIMING is not None:
        ev1.record()
        KERNEL_TIMING.append((_lib.lib().jg_last_kernel().decode() or _wgrad_kernel_name(nbatch, H, W, Cin, Cout, R, S, pad, stride, out_mode), ev0, ev1,
                              2.0 * nbatch * B * Ho * Wo * Cout * R * S * Cin, (nbatch, B, Ho, Wo, Cin, Cout, R, splitk),
                              2.0 * nbatch * (B * H * W * Cin + B * Ho * Wo * Cout) + 4.0 * R * S * Cin * Cout))


# Deferred small weight gradients (round 5): inside `with deferred_wgrads():` every weight-gradient launch of a backward (plain zero padding) is
# collected instead of issued, and the context's exit issues them as GROUPED launches (jg_conv2d_wgrad_tn_group, 16 problems per grid).  The
# SegFormer generator's backward has ~190 of them per cut_model step, each a 10 - 40 us launch on 8 - 64 workgroups; they only feed the gradient
# arena (fp32 atomics), nothing downstream in the backward reads them.  Their operands (dy, x) stay alive until the flush.
WGRAD_DEFER = None
WGRAD_GROUP = os.environ.get("JG_WGRAD_GROUP", "1") != "0"


_WGRAD_SIDE = None          # (stream, flush threshold) of the active deferred_wgrads context, or None: flush on the current stream at the exit


def flush_deferred_wgrads():
    """issue the collected problems as grouped launches: on the current stream, or -- inside `deferred_wgrads(stream=...)` -- on that stream,
    ordered behind everything enqueued on the current stream so far (their operands are complete), leaving the current stream free to go on
    with the input-gradient chain"""
    global WGRAD_DEFER
    todo, WGRAD_DEFER = WGRAD_DEFER, ([] if WGRAD_DEFER is not None else None)
    if not todo:
        return 0
    by_dt = {}
    for dt, a, keep in todo:
        by_dt.setdefault(dt, []).append(a)
    if os.environ.get("JG_WGRAD_DUMP"):        # dev: the problems of one flush (tools/wgrad_group_model.py reads this)
        with open(os.environ["JG_WGRAD_DUMP"], "a") as f:
            f.write("flush %d\n" % len(todo))
            for dt, a, keep in todo:
                f.write("%d %d %d %d %d %d %d %d %d %d %d %d %d\n" % (a.B, a.H, a.W, a.Cin, a.Cout, a.R, a.S, a.pad, a.stride, a.Ho, a.Wo, a.splitk, int(bool(a.dbias))))
    side = _WGRAD_SIDE[0] if _WGRAD_SIDE is not None else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        for dt, a, keep in todo:
            for t in keep[:2]:
                t.record_stream(side)              # dy / x come from the current stream's allocator pool
        ctx = torch.cuda.stream(side)
    else:
        import contextlib

        ctx = contextlib.nullcontext()
    with ctx:
        for dt, args in by_dt.items():
            arr = (WgradArgs * len(args))(*args)
            check(_lib.lib().jg_conv2d_wgrad_tn_group(dt, arr, len(args), _st()), "jg_conv2d_wgrad_tn_group")
    return len(todo)


class deferred_wgrads:
    """collect the weight-gradient launches of the enclosed backward and issue them grouped (JG_WGRAD_GROUP=0: off).  Without `stream`: one flush at
    the exit, on the current stream.  With `stream`: a flush onto that stream every `every` collected problems and at the exit, where the current
    stream joins it -- the grouped launches are bandwidth-bound full-chip kernels, the backward chain next to them is a string of small launches."""

    def __init__(self, stream=None, every=48):
        self.stream, self.every = stream, every

    def __enter__(self):
        global WGRAD_DEFER, _WGRAD_SIDE
        self.prev = WGRAD_DEFER
        if WGRAD_GROUP and WGRAD_DEFER is None:
            WGRAD_DEFER = []
            _WGRAD_SIDE = (self.stream, self.every) if self.stream is not None else None
        return self

    def __exit__(self, *exc):
        global WGRAD_DEFER, _WGRAD_SIDE
        if self.prev is None and WGRAD_DEFER is not None:
            try:
                if exc[0] is None:
                    flush_deferred_wgrads()
                    if _WGRAD_SIDE is not None:
                        torch.cuda.current_stream().wait_stream(_WGRAD_SIDE[0])
            finally:
                WGRAD_DEFER = None
                _WGRAD_SIDE = None


def axpby(a, alpha=1.0, b=None, beta=0.0, alpha_dev=None, out=None):
    y = torch.empty_like(a) if out is None else out
    check(_lib.lib().jg_axpby(_dt(a), a.data_ptr(), alpha, _p(alpha_dev), _p(b), beta, y.data_ptr(), a.numel(), _st()),
          "jg_axpby")
    return y


# ======================================================================================
# convolution
# ======================================================================================
class ConvMeta:
    """Geometry + working weight copies of one conv layer (filled by ParamArena)."""

    __slots__ = ("Cin", "Cout", "Cin_real", "Cout_real", "R", "S", "pad", "stride", "w16", "w16T", "weight", "bias",
                 "bias_pad")

    def out_hw(self, H, W):
        return ((H + 2 * self.pad - self.R) // self.stride + 1, (W + 2 * self.pad - self.S) // self.stride + 1)


WGRAD_TARGET_BLOCKS = int(os.environ.get("JG_WGRAD_TARGET_BLOCKS", "1536"))
WGRAD_MAX_SPLITK = 512  # deeper splits only add atomic traffic on a tiny output
WGRAD_SPLIT_PIX = int(os.environ.get("JG_WGRAD_SPLIT_PIX", "256"))     # fewest pixels a K slice of the weight gradient is worth


def _wgrad_splitk(tiles, mpix, nbatch=1):
    want = max(1, (WGRAD_TARGET_BLOCKS + tiles - 1) // tiles)
    cap = max(1, mpix // WGRAD_SPLIT_PIX)
    return max(1, min(want, cap, WGRAD_MAX_SPLITK, 65535 // max(1, nbatch)))


def conv2d_forward(x, m: ConvMeta, res=None, res_scale=1.0, alpha=1.0, stats=None):
    _require_cuda(x)
    B, H, W, Cin = x.shape
    assert Cin == m.Cin, (Cin, m.Cin)
    Ho, Wo = m.out_hw(H, W)
    y = torch.empty((B, Ho, Wo, m.Cout), device=x.device, dtype=x.dtype)
    conv_nt(x, m.w16, y, B=B, H=H, W=W, Cin=Cin, Cout=m.Cout, R=m.R, S=m.S, pad=m.pad, stride=m.stride, Ho=Ho, Wo=Wo,
            ldx=Cin, ldw=m.R * m.S * Cin, ldy=m.Cout, bias=m.bias_pad if m.bias_pad is not None else m.bias, res=res,
            ldres=m.Cout, alpha=alpha, res_scale=res_scale, stats=stats)
    return y


def dilate2d(x, Ho, Wo, stride):
    """zero insertion: y[:, s*i, s*j] = x[:, i, j] in a [B, Ho, Wo, C] buffer."""
    B, H, W, Cc = x.shape
    y = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_dilate2d(_dt(x), x.contiguous().data_ptr(), y.data_ptr(), B, H, W, Cc, Ho, Wo, stride, _st()), "jg_dilate2d")
    return y


# stride-2 transposed convolutions (input gradient of a stride-2 convolution, nn.ConvTranspose2d) in the four-phase form of the halo-resident
# kernel (jg_transposed_fold + jg_conv2d_nt x_mode 2) instead of a zero-dilated copy + a stride-1 convolution over it; JG_PHASE_TCONV=0: off
PHASE_TCONV = os.environ.get("JG_PHASE_TCONV", "1") != "0"
PATCH_DGRAD = os.environ.get("JG_PATCH_DGRAD", "1") != "0"      # round 6: input gradient of kernel == stride convolutions as one GEMM (conv2d_dgrad)


def phase_tconv_ok(m: ConvMeta, Hout, Wout, C, N):
    """shape limits of the phase form: kernel 3 or 4, stride 2, padding 1, output exactly twice the input, the halo kernel's channel / tile multiples"""
    return (PHASE_TCONV and m.stride == 2 and m.R == m.S and m.R in (3, 4) and m.pad == 1 and C % 64 == 0 and N % 64 == 0 and Hout % 32 == 0
            and Wout % 32 == 0)


def phase_tconv(x, m: ConvMeta, y, bias=None, alpha=1.0):
    """y [B, 2h, 2w, N] = transposed convolution of x [B, h, w, C] with the strided convolution's weights (m.w16T [N][R][S][C])"""
    B, h, w, C = x.shape
    N = y.shape[-1]
    wf = torch.empty((4, N, 2, 2, C), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_transposed_fold(_dt(x), m.w16T.data_ptr(), wf.data_ptr(), N, C, m.R, m.S, m.pad, _st()), "jg_transposed_fold")
    conv_nt(x, wf, y, B=B, H=2 * h, W=2 * w, Cin=C, Cout=N, R=3, S=3, pad=1, stride=1, Ho=2 * h, Wo=2 * w, ldx=C, ldw=4 * C, ldy=N, bias=bias,
            alpha=alpha, x_mode=2)
    return y


def conv2d_dgrad(dy, m: ConvMeta, x_shape, alpha=1.0):
    """dx = conv(dy, flipped/transposed weights).  Stride s > 1: the same stride-1 convolution over the zero-dilated
    dy (length H + 2p - k + 1 per axis) with pad k-1-p."""
    B, H, W, Cin = x_shape
    _, Ho, Wo, Cout = dy.shape
    dx = torch.empty(x_shape, device=dy.device, dtype=dy.dtype)
    if m.stride != 1:
        if H == 2 * Ho and W == 2 * Wo and phase_tconv_ok(m, H, W, Cout, Cin):
            return phase_tconv(dy, m, dx, alpha=alpha)
        if (PATCH_DGRAD and Cin == 8 and m.Cin_real <= 4 and Cout % 8 == 0 and m.R * m.S * Cout * 16 <= 65536
                and Ho == (H + 2 * m.pad - m.R) // m.stride + 1 and Wo == (W + 2 * m.pad - m.S) // m.stride + 1):
            # onto an image (<= 4 real channels): gather form, s^2 x fewer multiply-adds than the dilated convolution and no MFMA padding of 3 -> 8
            check(_lib.lib().jg_conv_dgrad_gather(_dt(dy), dy.contiguous().data_ptr(), m.w16.data_ptr(), dx.data_ptr(), B, H, W, Ho, Wo, Cout, m.R, m.S,
                                                  m.stride, m.pad, alpha, _st()), "jg_conv_dgrad_gather")
            return dx
        if PATCH_DGRAD and m.stride == m.R == m.S and m.pad == 0 and H == Ho * m.R and W == Wo * m.S:
            # kernel == stride (the spatial-reduction convolutions of the MiT attention, segformer/backbone.py): the patches do not overlap, the
            # input gradient of a patch is dy . W -- ONE GEMM with R S Cin output columns per output pixel + a depth-to-space copy.  (The dilated
            # form below runs an R x S convolution over a tensor of which one tap in R S is non-zero: 64 x the MFMA work at sr_ratio 8.)
            N = m.R * m.S * Cin
            wt = m.w16.view(Cout, N).t().contiguous()                          # [(ky, kx, ci)][co]
            tmp = torch.empty((B, Ho, Wo, N), device=dy.device, dtype=dy.dtype)
            conv_nt(dy, wt, tmp, B=B, H=Ho, W=Wo, Cin=Cout, Cout=N, R=1, S=1, pad=0, stride=1, Ho=Ho, Wo=Wo, ldx=Cout, ldw=Cout, ldy=N, alpha=alpha)
            dx.view(B, Ho, m.R, Wo, m.S, Cin).copy_(tmp.view(B, Ho, Wo, m.R, m.S, Cin).permute(0, 1, 3, 2, 4, 5))
            return dx
        Hd, Wd = H + 2 * m.pad - m.R + 1, W + 2 * m.pad - m.S + 1
        dyd = dilate2d(dy, Hd, Wd, m.stride)
        conv_nt(dyd, m.w16T, dx, B=B, H=Hd, W=Wd, Cin=Cout, Cout=Cin, R=m.R, S=m.S, pad=m.R - 1 - m.pad, stride=1, Ho=H, Wo=W,
                ldx=Cout, ldw=m.R * m.S * Cout, ldy=Cin, alpha=alpha)
        return dx
    conv_nt(dy, m.w16T, dx, B=B, H=Ho, W=Wo, Cin=Cout, Cout=Cin, R=m.R, S=m.S, pad=m.R - 1 - m.pad, stride=1, Ho=H, Wo=W,
            ldx=Cout, ldw=m.R * m.S * Cout, ldy=Cin, alpha=alpha)
    return dx


def conv2d_wgrad(dy, x, m: ConvMeta, alpha=1.0, want_w=True, want_b=True):
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    wg = m.weight.grad
    if wg is None:
        raise RuntimeError("conv weight has no arena-backed .grad (module not finalised by ParamArena)")
    ktot = m.R * m.S * Cin
    tiles = ((Cout + 127) // 128) * ((ktot + 127) // 128)  # Cout <= 64 runs the 64-row tile: same count
    splitk = _wgrad_splitk(tiles, B * Ho * Wo)
    dbias = m.bias.grad if (want_b and m.bias is not None) else None
    wgrad_tn(dy, x, wg, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=m.R, S=m.S, pad=m.pad, stride=m.stride, Ho=Ho, Wo=Wo,
             lddy=Cout, ldx=Cin, lddw=m.R * m.S * m.Cin_real, dbias=dbias, Cin_out=m.Cin_real, Cout_out=m.Cout_real,
             splitk=splitk, alpha=alpha)


# ---- row-packed 7x7 head (round 6) -----------------------------------------------------------------------------------------------------------
HEAD7_PACKED = os.environ.get("JG_HEAD7_PACKED", "1") != "0"
_HEAD7_BUFS = {}      # per convolution (keyed by its 16-bit weight copy): the packed weight buffers of _Head7x7Fn


def head7_ok(x_pad, m: ConvMeta):
    """shapes the row-packed form serves: 7x7, stride 1, no padding of its own (the caller has reflect-padded), 64 k input channels, <= 4 real
    output channels, a map the halo-resident kernels tile (W a multiple of 16)"""
    if not HEAD7_PACKED or TORCH_OPS_BOUNDARY or KERNEL_TIMING is not None or not x_pad.is_cuda:
        return False
    B, Hp, Wp, Cin = x_pad.shape
    return (m.R == 7 and m.S == 7 and m.stride == 1 and m.pad == 0 and Cin == m.Cin and Cin % 64 == 0 and m.Cout_real <= 4 and (Wp - 6) % 16 == 0
            and Hp > 6 and B * (Hp - 6) * (Wp - 6) >= 65536 and x_pad.dtype in (torch.float16, torch.bfloat16))


class _Head7x7Fn(JGFunction):
    """act(conv7x7(x_pad) + b) for <= 4 output channels, row-packed (csrc/elementwise.hip tapsum7): a 1 x 7 convolution onto 7 x 4 packed
    channels on the halo-resident kernel, then the sum over the tap rows -- 7 x fewer multiply-adds than the 7x7 convolution whose 3 output
    channels are padded to an MFMA tile of 32; the backward is the same two steps transposed."""

    @staticmethod
    def forward(ctx, x_pad, weight, bias, meta, act):
        m = meta
        x_pad = x_pad.contiguous()
        B, Hp, Wp, Cin = x_pad.shape
        H, W = Hp - 6, Wp - 6
        dev, dt = x_pad.device, x_pad.dtype
        # packed weight copies, rebuilt from the arena's 16-bit copy on every forward (two small launches; the buffers persist with the module so
        # that a captured graph replays into the same addresses): wz [32 = (ky, c)][1][7][Cin] for the forward stage and the weight
        # gradient's layout, wzT [Cin][1][7 flipped][32] for the input-gradient stage
        key = (m.w16.data_ptr(), dt, Cin)
        bufs = _HEAD7_BUFS.get(key)
        if bufs is None:
            bufs = _HEAD7_BUFS[key] = (torch.zeros((8, 4, 7, Cin), device=dev, dtype=dt), torch.zeros((Cin, 7, 8, 4), device=dev, dtype=dt))
        wz, wzT = bufs
        wz[:7].copy_(m.w16[:4].permute(1, 0, 2, 3))
        wzT[:, :, :7].copy_(m.w16[:4].permute(3, 2, 1, 0).flip(1))              # [ci][6 - kx][ky][c]
        z = torch.empty((B, Hp, W, 32), device=dev, dtype=dt)
        conv_nt(x_pad, wz, z, B=B, H=Hp, W=Wp, Cin=Cin, Cout=32, R=1, S=7, pad=0, stride=1, Ho=Hp, Wo=W, ldx=Cin, ldw=7 * Cin, ldy=32)
        out = torch.empty((B, H, W, 8), device=dev, dtype=dt)
        b4 = m.bias_pad                                                        # fp32, padded to the 8-channel pixel (None: no bias)
        if b4 is None and m.bias is not None:
            b4 = torch.zeros(4, device=dev, dtype=torch.float32)
            b4[:m.Cout_real].copy_(m.bias.detach()[:m.Cout_real])
        check(_lib.lib().jg_tapsum7(_dt(x_pad), z.data_ptr(), _p(b4), out.data_ptr(), B, H, W, act, _st()), "jg_tapsum7")
        ctx.save_for_backward(x_pad, out, wzT)
        ctx.cfg = (m, act)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x_pad, out, wzT = ctx.saved_tensors
        m, act = ctx.cfg
        dout = dout.contiguous()
        B, Hp, Wp, Cin = x_pad.shape
        H, W = Hp - 6, Wp - 6
        dev, dt = x_pad.device, x_pad.dtype
        Hz = (Hp + 7) // 8 * 8
        dz = torch.empty((B, Hz, W, 32), device=dev, dtype=dt)
        dzm = torch.empty((B, Hp, W + 12, 32), device=dev, dtype=dt)
        check(_lib.lib().jg_tapspread7(_dt(dout), dout.data_ptr(), out.data_ptr(), dz.data_ptr(), dzm.data_ptr(), B, H, W, Hz, act, _st()), "jg_tapspread7")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x_pad)
            conv_nt(dzm, wzT, dx, B=B, H=Hp, W=W + 12, Cin=32, Cout=Cin, R=1, S=7, pad=0, stride=1, Ho=Hp, Wo=Wp, ldx=32, ldw=7 * 32, ldy=Cin)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            wg = m.weight.grad
            if wg is None:
                raise RuntimeError("conv weight has no arena-backed .grad (module not finalised by ParamArena)")
            acc = torch.zeros(32 * 7 * Cin + 32, device=dev, dtype=torch.float32)          # packed weight gradient + bias sums, one clear
            dwz, db = acc[:32 * 7 * Cin].view(8, 4, 7, Cin), acc[32 * 7 * Cin:]
            wgrad_tn(dz, x_pad, dwz, B=B, H=Hp, W=Wp, Cin=Cin, Cout=32, R=1, S=7, pad=0, stride=1, Ho=Hz, Wo=W, lddy=32, ldx=Cin, lddw=7 * Cin, dbias=db,
                     Cin_out=Cin, Cout_out=32, splitk=1, defer=False)
            cr = m.Cout_real
            wg.permute(0, 2, 3, 1)[:cr].add_(dwz[:7, :cr].permute(1, 0, 2, 3)[..., :m.Cin_real])      # physical [Cout][R][S][Cin]
            if m.bias is not None and ctx.needs_input_grad[2]:
                m.bias.grad[:cr].add_(db[:cr])
        return dx, None, None, None, None


def head_conv7(x_pad, m: ConvMeta, act=0):
    """ReflectionPad2d(3)'s output -> act(Conv2d(Cin, <= 4, 7)(x_pad)) [B, H, W, 8]; callers check head7_ok first"""
    return _Head7x7Fn.apply(x_pad, m.weight, m.bias, m, act)


class _Conv2dFn(JGFunction):
    @staticmethod
    def forward(ctx, x, weight, bias, res, meta, res_scale, alpha, stats=None):
        # stats (round 6, conv2d_stats): a zeroed fp32 [B, Cout, 2] buffer the launch's epilogue fills with the per-(image, channel) sum and sum
        # of squares of its output -- the statistics pass of the InstanceNorm / GroupNorm that follows (not a differentiable input)
        y = conv2d_forward(x, meta, res, res_scale, alpha, stats)
        ctx.save_for_backward(x)
        ctx.meta, ctx.res_scale, ctx.alpha, ctx.has_res = meta, res_scale, alpha, res is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        m = ctx.meta
        dy = dy.contiguous()
        dx = dres = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_dgrad(dy, m, x.shape, ctx.alpha)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            conv2d_wgrad(dy, x, m, ctx.alpha, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy if ctx.res_scale == 1.0 else axpby(dy, ctx.res_scale)
        return dx, None, None, dres, None, None, None, None


# torch.ops boundary mode: inside `with torch_ops_boundary():` every op of the palette step -- convolutions, norms,
# attention, resampling, the embedding MLPs, q_sample and the loss -- is a call of `torch.ops.jg355.*` (torch.library custom ops with
# fake kernels and autograd formulas) on the fp32 master parameters; concatenations / scalings are plain ATen ops and autograd assembles
# the backward.  Same kernels through the same C ABI; what it gives up is what the arena-accumulating nodes and the fused UNet schedule
# exist for (no gradient copies, one autograd node).  tests/test_gpu_1_model.py::test_palette_step_through_torch_ops pins the two forms.
TORCH_OPS_BOUNDARY = False


class torch_ops_boundary:
    def __enter__(self):
        global TORCH_OPS_BOUNDARY
        self.prev, TORCH_OPS_BOUNDARY = TORCH_OPS_BOUNDARY, True

    def __exit__(self, *a):
        global TORCH_OPS_BOUNDARY
        TORCH_OPS_BOUNDARY = self.prev


def _conv2d_via_torch_ops(x, m: ConvMeta, res, res_scale, alpha):
    """the layer on torch.ops.jg355.conv2d_nt with its fp32 MASTER weight as a differentiable input ([Cout, R, S, Cin], zero-padded to
    the 8-channel granularity of the activations like the arena's working copy)"""
    import torch.nn.functional as F

    w = m.weight.reshape(m.Cout_real, m.Cin_real, m.R, m.S).permute(0, 2, 3, 1)
    b = m.bias
    if m.Cin != m.Cin_real or m.Cout != m.Cout_real:
        w = F.pad(w, (0, m.Cin - m.Cin_real, 0, 0, 0, 0, 0, m.Cout - m.Cout_real))
        if b is not None:
            b = F.pad(b, (0, m.Cout - m.Cout_real))
    return torch.ops.jg355.conv2d_nt(x, w, b, res, m.pad, m.stride, alpha, res_scale if res is not None else 0.0)


# round 6, A/B: InstanceNorm statistics of the mobile ResNet blocks in the epilogue of the point-wise convolution in front (jg_conv_args.stats, as the
# fused UNet schedule does for its 3x3 layers).  MEASURED SLOWER: mobile_resnet_attn + [projected_d, basic] 45.0 -> 49.3 ms -- on a 45 us GEMM-shaped
# launch the LDS-transposed reduction + 256 fp32 atomics per tile cost more than the 27 us statistics pass they replace.  Off.
CONV_STATS = os.environ.get("JG_CONV_STATS", "0") != "0"


def conv2d_stats(x, meta: ConvMeta):
    """(conv2d(x), sums): sums fp32 [B, Cout, 2] = per-(image, channel) sum / sum of squares of the output, accumulated by the convolution's
    epilogue (jg_conv_args.stats) -- hand it to group_norm(..., sums=sums) and the norm skips its statistics pass; None where the epilogue form
    does not exist (shape limits of jg_conv2d_nt's stats: whole 256-pixel tiles inside an image, 64-channel multiples; op boundary; timing runs)"""
    B, H, W, _ = x.shape
    Ho, Wo = meta.out_hw(H, W)
    if (not CONV_STATS or TORCH_OPS_BOUNDARY or KERNEL_TIMING is not None or (Ho * Wo) % 256 or meta.Cout % 64 or meta.R != meta.S or meta.R not in (1, 3)
            or meta.stride != 1 or _lib.lib().jg_get_tuning(b"JG_DETERMINISTIC") != 0):
        return conv2d(x, meta), None
    sums = torch.zeros((B, meta.Cout, 2), device=x.device, dtype=torch.float32)
    return _Conv2dFn.apply(x, meta.weight, meta.bias, None, meta, 1.0, 1.0, sums), sums


def conv2d(x, meta: ConvMeta, res=None, res_scale=1.0, alpha=1.0):
    """y = alpha*conv(x) + bias + res_scale*res   (nn.Conv2d / Conv1d(k=1) of the reference)."""
    if TORCH_OPS_BOUNDARY:
        return _conv2d_via_torch_ops(x, meta, res, res_scale, alpha)
    return _Conv2dFn.apply(x, meta.weight, meta.bias, res, meta, res_scale, alpha)


# input gradient of the reflect-padded 3x3 convolutions on the halo-resident kernel + a ring kernel (1, round 6) or as a full convolution over
# the padded domain on the im2col kernel + reflect_pad_bwd (0: rounds 1-5)
REFLECT_DGRAD_HALO = os.environ.get("JG_REFLECT_DGRAD_HALO", "1") != "0"


def reflect_conv_ok(x, m: ConvMeta):
    """shape limits of pad_mode = 1 (halo-resident kernels): ReflectionPad2d(1) + 3x3 / stride 1 / pad 0 convolution"""
    B, H, W, Cin = x.shape
    return (m.R == 3 and m.S == 3 and m.stride == 1 and m.pad == 0 and Cin % 64 == 0 and m.Cout % 64 == 0 and H % 16 == 0
            and W % 16 == 0 and H >= 16 and W >= 16 and B * H * W * max(Cin, m.Cout) < (1 << 31))


class _ReflectConv2dFn(JGFunction):
    """nn.ReflectionPad2d(1) -> nn.Conv2d(3x3, padding 0) as one launch: the halo-resident kernel mirrors the border pixels while
    it loads its halo, so the padded tensor is never written.  Backward: the weight gradient reads x with the same mirrored halo;
    the input gradient is the full convolution over the (H+2, W+2) padded domain folded back by the reflection's adjoint."""

    @staticmethod
    def forward(ctx, x, weight, bias, meta, with_identity=False):
        _require_cuda(x)
        x = x.contiguous()
        B, H, W, Cin = x.shape
        m = meta
        y = torch.empty((B, H, W, m.Cout), device=x.device, dtype=x.dtype)
        conv_nt(x, m.w16, y, B=B, H=H, W=W, Cin=Cin, Cout=m.Cout, R=3, S=3, pad=1, stride=1, Ho=H, Wo=W, ldx=Cin, ldw=9 * Cin,
                ldy=m.Cout, bias=m.bias_pad if m.bias_pad is not None else m.bias, pad_mode=1)
        ctx.save_for_backward(x)
        ctx.meta = m
        ctx.with_identity = with_identity
        if with_identity:          # (x, conv(x)): the gradient of the residual branch `x + f(x)` comes back through the first output
            return x.view_as(x), y
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        (x,) = ctx.saved_tensors
        m = ctx.meta
        did, dy = grads if ctx.with_identity else (None, grads[0])
        if dy is None:
            return did, None, None, None, None
        dy = dy.contiguous()
        B, H, W, Cin = x.shape
        dx = None
        if ctx.needs_input_grad[0]:
            if REFLECT_DGRAD_HALO and m.Cout % 64 == 0 and Cin % 64 == 0:
                # round 6: the interior of the padded-domain gradient IS the zero-padded input gradient on H x W (halo-resident kernel); the
                # one-pixel ring of the reflection's adjoint is added by jg_reflect_dgrad_border (csrc/reflect_border.hip).  The residual
                # branch's gradient (with_identity) rides in as the residual addend of that convolution: no accumulation launch of autograd's
                dx = torch.empty_like(x)
                res = did.contiguous() if did is not None else None
                did = None
                conv_nt(dy, m.w16T, dx, B=B, H=H, W=W, Cin=m.Cout, Cout=Cin, R=3, S=3, pad=1, stride=1, Ho=H, Wo=W, ldx=m.Cout, ldw=9 * m.Cout,
                        ldy=Cin, res=res, ldres=Cin)
                ws = torch.empty(_lib.lib().jg_reflect_dgrad_border_ws_floats(B, H, W, Cin), device=x.device, dtype=torch.float32)
                check(_lib.lib().jg_reflect_dgrad_border(_dt(dy), dy.data_ptr(), m.Cout, m.w16T.data_ptr(), dx.data_ptr(), Cin, ws.data_ptr(), B, H, W,
                                                         m.Cout, Cin, 1.0, _st()), "jg_reflect_dgrad_border")
            else:
                dxp = conv2d_dgrad(dy, m, (B, H + 2, W + 2, Cin))
                dx = torch.empty_like(x)
                check(_lib.lib().jg_reflect_pad2d_bwd(_dt(dy), dxp.data_ptr(), dx.data_ptr(), B, H, W, Cin, 1, _st()), "jg_reflect_pad2d_bwd")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            wg = m.weight.grad
            if wg is None:
                raise RuntimeError("conv weight has no arena-backed .grad (module not finalised by ParamArena)")
            dbias = m.bias.grad if (ctx.needs_input_grad[2] and m.bias is not None) else None
            wgrad_tn(dy, x, wg, B=B, H=H, W=W, Cin=Cin, Cout=m.Cout, R=3, S=3, pad=1, stride=1, Ho=H, Wo=W, lddy=m.Cout, ldx=Cin,
                     lddw=9 * m.Cin_real, dbias=dbias, Cin_out=m.Cin_real, Cout_out=m.Cout_real,
                     splitk=_wgrad_splitk(((m.Cout + 127) // 128) * ((9 * Cin + 127) // 128), B * H * W), pad_mode=1)
        if did is not None and dx is not None:          # (the fold path: no residual epilogue there)
            dx = axpby(dx, 1.0, did.contiguous(), 1.0)
        elif did is not None:
            dx = did
        return dx, None, None, None, None


def reflect_conv2d_id(x, meta: ConvMeta):
    """(x, conv(reflection_pad(x, 1))): use the first output as the identity of `x + f(x)` -- its gradient is added inside the input-gradient
    convolution of this layer (round 6) instead of by autograd's accumulation kernel"""
    return _ReflectConv2dFn.apply(x, meta.weight, meta.bias, meta, True)


def reflect_conv2d(x, meta: ConvMeta):
    """conv(reflection_pad(x, 1)) for a 3x3 / pad-0 ConvMeta (see reflect_conv_ok)."""
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.reflect_conv2d(x, meta.weight.permute(0, 2, 3, 1), meta.bias)
    return _ReflectConv2dFn.apply(x, meta.weight, meta.bias, meta)


# ======================================================================================
# GroupNorm (+FiLM, +SiLU) / InstanceNorm1d
# ======================================================================================
# Zeroed fp32 scratch for the accumulating reductions of the norm nodes (round 6): jg_gn_stats / jg_gn_bwd_reduce clear their [B, C, 2] rows with a
# hipMemsetAsync of their own -- one 5 us launch per norm and direction, 245 per step of the mobile ResNet generator.  The rows now come out of a
# chunk that ONE torch.zeros clears (4 MB, ~60 norms); a chunk lives as long as any row handed out of it.
ZERO_POOL = os.environ.get("JG_ZERO_POOL", "1") != "0"
_ZERO_CHUNK = {}


def zero_pool_reset(device=None, allocate=False):
    """forget the current chunk(s); allocate: start a fresh one NOW on the current stream.  A hipGraph capture calls this as its first action
    (allocate=True: the chunk's clear becomes a node of THAT graph and is replayed with it) and again after the capture (the eager code must
    not take rows out of a captured chunk, whose clear it does not replay)."""
    for k in [k for k in _ZERO_CHUNK if device is None or k[0] == device]:
        del _ZERO_CHUNK[k]
    if allocate and ZERO_POOL and device is not None:
        _ZERO_CHUNK[(device, torch.cuda.current_stream(device).cuda_stream)] = [torch.zeros(1 << 20, device=device, dtype=torch.float32), 0,
                                                                               torch.cuda.is_current_stream_capturing()]


def zeros_f32(n, device):
    """n zeroed fp32 values (a view of a pooled chunk; never written by anyone else).  Chunks are per STREAM: a row is used on the stream
    that enqueued its chunk's clear (a forked branch of a capture gets a chunk of its own, cleared on that branch).  Inside a capture that did
    not prepare a chunk (zero_pool_reset) every request is a torch.zeros of its own: a row is only ever cleared by a launch of the graph
    that accumulates into it."""
    n = (int(n) + 63) // 64 * 64
    if not ZERO_POOL or n > (1 << 18) or device.type != "cuda":
        return torch.zeros(n, device=device, dtype=torch.float32)
    cap = torch.cuda.is_current_stream_capturing()
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    st = _ZERO_CHUNK.get(key)
    if st is not None and st[2] != cap:
        st = None
    if st is None and cap and not any(v[2] for v in _ZERO_CHUNK.values()):
        return torch.zeros(n, device=device, dtype=torch.float32)          # a capture that did not opt in
    if st is None or st[1] + n > st[0].numel():
        st = _ZERO_CHUNK[key] = [torch.zeros(1 << 20, device=device, dtype=torch.float32), 0, cap]
    out = st[0][st[1]:st[1] + n]
    st[1] += n
    return out


class _GroupNormFn(JGFunction):
    @staticmethod
    def forward(ctx, x, gamma, beta, film, G, act, eps, sums=None, add=None):
        _require_cuda(x)
        L = _lib.lib()
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        dev, st, dt = x.device, _st(), _dt(x)
        ab = torch.empty((B, C, 2), device=dev, dtype=torch.float32)
        mr = torch.empty((B, G, 2), device=dev, dtype=torch.float32)
        y = torch.empty_like(x)
        ldfilm = film.stride(0) if film is not None else 0
        if sums is None:            # (given: the producing convolution's epilogue has accumulated them, ops.conv2d_stats)
            sums = zeros_f32(B * C * 2, dev)[:B * C * 2].view(B, C, 2)
            check(L.jg_gn_stats_ld(dt, x.data_ptr(), C, sums.data_ptr(), C, B, HW, C, st), "jg_gn_stats_ld")
        check(L.jg_gn_coef(sums.data_ptr(), _p(gamma), _p(beta), _p(film), ldfilm, ab.data_ptr(), mr.data_ptr(), B, HW, C,
                           G, eps, st), "jg_gn_coef")
        if add is not None:          # y = act(norm(x)) + add in the apply pass (round 6); d(add) = dy
            add = add.contiguous()
            check(L.jg_gn_apply_add(dt, x.data_ptr(), C, ab.data_ptr(), add.data_ptr(), C, y.data_ptr(), C, B, HW, C, act, st), "jg_gn_apply_add")
        else:
            check(L.jg_gn_apply(dt, x.data_ptr(), ab.data_ptr(), y.data_ptr(), B, HW, C, act, st), "jg_gn_apply")
        ctx.save_for_backward(x, ab, mr, gamma, beta, film)
        ctx.G, ctx.act = G, act
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, ab, mr, gamma, beta, film = ctx.saved_tensors
        L = _lib.lib()
        dy = dy.contiguous()
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        dev, st, dt = x.device, _st(), _dt(x)
        red = zeros_f32(B * C * 2, dev)[:B * C * 2].view(B, C, 2)
        pqr = torch.empty((B, C, 3), device=dev, dtype=torch.float32)
        want_film = film is not None and ctx.needs_input_grad[3]
        dfilm = torch.empty((B, 2 * C), device=dev, dtype=torch.float32) if want_film else None
        dgamma = gamma.grad if (gamma is not None and ctx.needs_input_grad[1]) else None
        dbeta = beta.grad if (beta is not None and ctx.needs_input_grad[2]) else None
        if (gamma is not None and ctx.needs_input_grad[1] and dgamma is None):
            raise RuntimeError("norm weight has no arena-backed .grad")
        ldfilm = film.stride(0) if film is not None else 0
        check(L.jg_gn_bwd_reduce_ld_acc(dt, x.data_ptr(), C, dy.data_ptr(), C, ab.data_ptr(), red.data_ptr(), B, HW, C, ctx.act, st),
              "jg_gn_bwd_reduce_ld_acc")
        check(L.jg_gn_bwd_coef(red.data_ptr(), _p(gamma), _p(beta), _p(film), ldfilm, mr.data_ptr(), pqr.data_ptr(),
                               _p(dgamma), _p(dbeta), _p(dfilm), 2 * C, B, HW, C, ctx.G, st), "jg_gn_bwd_coef")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(L.jg_gn_bwd_apply(dt, x.data_ptr(), dy.data_ptr(), ab.data_ptr(), pqr.data_ptr(), dx.data_ptr(), B, HW,
                                    C, ctx.act, st), "jg_gn_bwd_apply")
        return dx, None, None, dfilm, None, None, None, None, (dy if ctx.needs_input_grad[8] else None)


_GN_STATUS = {}


def gn_status(device):
    """the sticky error word of jg_gn_bwd_fused (one per device): non-zero = an inter-workgroup wait of a fused GroupNorm backward
    expired (that launch's numbers are invalid).  `check_gn_status()` reads it (host synchronisation: tests, bench, smoke)."""
    key = torch.device(device).index or 0
    t = _GN_STATUS.get(key)
    if t is None:
        t = _GN_STATUS[key] = torch.zeros(4, device=device, dtype=torch.int32)
    return t


def check_gn_status():
    for k, t in _GN_STATUS.items():
        if int(t[0]) != 0:
            raise RuntimeError(f"jg_gn_bwd_fused: an inter-workgroup wait expired on cuda:{k} (results of that launch are invalid)")


def group_norm(x, G, gamma=None, beta=None, film=None, act=JG_ACT_NONE, eps=1e-5, sums=None, add=None):
    """act(GroupNorm_G(x) * gamma + beta [* (1 + scale) + shift]); statistics in fp32.  sums: the [B, C, 2] statistics of x where a
    convolution's epilogue has already taken them (conv2d_stats)."""
    if TORCH_OPS_BOUNDARY:
        y = torch.ops.jg355.group_norm_act(x, gamma, beta, film, G, act, eps)
        return y if add is None else y + add
    return _GroupNormFn.apply(x, gamma, beta, film, G, act, eps, sums, add)


# ======================================================================================
# CUT networks: transposed convolution, reflection padding, stand-alone activations
# ======================================================================================
def conv_transpose2d_forward(x, m: ConvMeta, output_padding=0):
    """nn.ConvTranspose2d(Cin_t, Cout_t, k, stride s, padding p, output_padding op).  `m` holds the weight as the
    stride-s CONV whose input-gradient this is (m.Cout = Cin_t, m.Cin = Cout_t): forward = stride-1 conv of the
    zero-dilated input with m.w16T (the flipped / transposed copy)."""
    B, H, W, Cin_t = x.shape
    assert Cin_t == m.Cout, (Cin_t, m.Cout)
    Ho = (H - 1) * m.stride - 2 * m.pad + m.R + output_padding
    Wo = (W - 1) * m.stride - 2 * m.pad + m.S + output_padding
    y = torch.empty((B, Ho, Wo, m.Cin), device=x.device, dtype=x.dtype)
    if Ho == 2 * H and Wo == 2 * W and phase_tconv_ok(m, Ho, Wo, Cin_t, m.Cin):
        return phase_tconv(x.contiguous(), m, y, bias=m.bias)
    Hd, Wd = Ho + 2 * m.pad - m.R + 1, Wo + 2 * m.pad - m.S + 1
    xd = dilate2d(x, Hd, Wd, m.stride)
    conv_nt(xd, m.w16T, y, B=B, H=Hd, W=Wd, Cin=Cin_t, Cout=m.Cin, R=m.R, S=m.S, pad=m.R - 1 - m.pad, stride=1, Ho=Ho, Wo=Wo,
            ldx=Cin_t, ldw=m.R * m.S * Cin_t, ldy=m.Cin, bias=m.bias)
    return y


class _ConvTranspose2dFn(JGFunction):
    @staticmethod
    def forward(ctx, x, weight, bias, meta, output_padding):
        _require_cuda(x)
        y = conv_transpose2d_forward(x, meta, output_padding)
        ctx.save_for_backward(x)
        ctx.meta = meta
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        m = ctx.meta
        dy = dy.contiguous()
        B, H, W, Cin_t = x.shape
        _, Ho, Wo, Cout_t = dy.shape
        dx = None
        if ctx.needs_input_grad[0]:      # the forward of the stride-s convolution
            dx = torch.empty_like(x)
            conv_nt(dy, m.w16, dx, B=B, H=Ho, W=Wo, Cin=Cout_t, Cout=Cin_t, R=m.R, S=m.S, pad=m.pad, stride=m.stride, Ho=H, Wo=W,
                    ldx=Cout_t, ldw=m.R * m.S * Cout_t, ldy=Cin_t)
        if ctx.needs_input_grad[1]:
            wg = m.weight.grad
            if wg is None:
                raise RuntimeError("conv-transpose weight has no arena-backed .grad")
            ktot = m.R * m.S * Cout_t
            tiles = ((Cin_t + 127) // 128) * ((ktot + 127) // 128)
            wgrad_tn(x, dy, wg, B=B, H=Ho, W=Wo, Cin=Cout_t, Cout=Cin_t, R=m.R, S=m.S, pad=m.pad, stride=m.stride, Ho=H, Wo=W,
                     lddy=Cin_t, ldx=Cout_t, lddw=ktot, splitk=_wgrad_splitk(tiles, B * H * W))
        if ctx.needs_input_grad[2] and m.bias is not None:
            check(_lib.lib().jg_channel_sum(_dt(dy), dy.data_ptr(), Cout_t, m.bias.grad.data_ptr(), B * Ho * Wo, Cout_t, 1.0, _st()),
                  "jg_channel_sum")
        return dx, None, None, None, None


def conv_transpose2d(x, meta: ConvMeta, output_padding=0):
    if TORCH_OPS_BOUNDARY:
        from .ops_library_cut import conv_transpose2d_via_ops

        return conv_transpose2d_via_ops(x, meta, output_padding)
    return _ConvTranspose2dFn.apply(x, meta.weight, meta.bias, meta, output_padding)


class _ReflectPadFn(JGFunction):
    @staticmethod
    def forward(ctx, x, pad):
        _require_cuda(x)
        B, H, W, Cc = x.shape
        y = torch.empty((B, H + 2 * pad, W + 2 * pad, Cc), device=x.device, dtype=x.dtype)
        check(_lib.lib().jg_reflect_pad2d(_dt(x), x.contiguous().data_ptr(), y.data_ptr(), B, H, W, Cc, pad, _st()), "jg_reflect_pad2d")
        ctx.pad, ctx.shape = pad, x.shape
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        B, H, W, Cc = ctx.shape
        dx = torch.empty(ctx.shape, device=dy.device, dtype=dy.dtype)
        check(_lib.lib().jg_reflect_pad2d_bwd(_dt(dy), dy.contiguous().data_ptr(), dx.data_ptr(), B, H, W, Cc, ctx.pad, _st()),
              "jg_reflect_pad2d_bwd")
        return dx, None


def reflect_pad2d(x, pad):
    """nn.ReflectionPad2d(pad) on NHWC."""
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.reflect_pad2d(x, pad)
    return _ReflectPadFn.apply(x, pad)


class _CropFn(JGFunction):
    @staticmethod
    def forward(ctx, x, top, left, Ho, Wo):
        _require_cuda(x)
        x = x.contiguous()
        B, H, W, Cc = x.shape
        y = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=x.dtype)
        check(_lib.lib().jg_crop2d(_dt(x), x.data_ptr(), y.data_ptr(), B, H, W, Cc, top, left, Ho, Wo, 0, _st()), "jg_crop2d")
        ctx.cfg = (H, W, top, left)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        H, W, top, left = ctx.cfg
        dy = dy.contiguous()
        B, Ho, Wo, Cc = dy.shape
        dx = torch.empty(B, H, W, Cc, device=dy.device, dtype=dy.dtype)
        check(_lib.lib().jg_crop2d(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, H, W, Cc, top, left, Ho, Wo, 1, _st()), "jg_crop2d")
        return dx, None, None, None, None


def crop2d(x, top, left, Ho, Wo):
    """x[:, top:top+Ho, left:left+Wo] on NHWC (adjoint: zero-filled placement)."""
    return _CropFn.apply(x, top, left, Ho, Wo)


class _ActFn(JGFunction):
    @staticmethod
    def forward(ctx, x, act):
        _require_cuda(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        check(_lib.lib().jg_act_fwd(_dt(x), x.data_ptr(), y.data_ptr(), x.numel(), act, _st()), "jg_act_fwd")
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(_lib.lib().jg_act_bwd(_dt(y), y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), ctx.act, _st()), "jg_act_bwd")
        return dx, None


def activation(x, act):
    """stand-alone nn.ReLU / nn.LeakyReLU(0.2) / nn.Tanh (JG_ACT_*) where no normalisation precedes the activation."""
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.act(x, act)
    return _ActFn.apply(x, act)


# ======================================================================================
# resampling / concat
# ======================================================================================
def _pool(x, scale):
    B, H, W, Cc = x.shape
    y = torch.empty((B, H // 2, W // 2, Cc), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_pool2x2(_dt(x), x.data_ptr(), y.data_ptr(), B, H, W, Cc, scale, _st()), "jg_pool2x2")
    return y


def _up(x, scale):
    B, H, W, Cc = x.shape
    y = torch.empty((B, H * 2, W * 2, Cc), device=x.device, dtype=x.dtype)
    check(_lib.lib().jg_upsample2x(_dt(x), x.data_ptr(), y.data_ptr(), B, H, W, Cc, scale, _st()), "jg_upsample2x")
    return y


class _AvgPool2Fn(JGFunction):
    @staticmethod
    def forward(ctx, x):
        return _pool(x, 0.25)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        return _up(dy.contiguous(), 0.25)


class _Upsample2Fn(JGFunction):
    @staticmethod
    def forward(ctx, x):
        return _up(x, 1.0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        return _pool(dy.contiguous(), 1.0)


def avg_pool2(x):
    return torch.ops.jg355.resample2(x, False, 0.25)


def upsample_nearest2(x):
    return torch.ops.jg355.resample2(x, True, 1.0)


def copy_channels(src, soff, dst, doff, n):
    P = src.numel() // src.shape[-1]
    check(_lib.lib().jg_copy_channels(_dt(src), src.data_ptr(), src.shape[-1], soff, dst.data_ptr(), dst.shape[-1], doff, P,
                                      n, _st()), "jg_copy_channels")


class _CatFn(JGFunction):
    @staticmethod
    def forward(ctx, a, b):
        Ca, Cb = a.shape[-1], b.shape[-1]
        y = torch.empty(a.shape[:-1] + (Ca + Cb,), device=a.device, dtype=a.dtype)
        copy_channels(a, 0, y, 0, Ca)
        copy_channels(b, 0, y, Ca, Cb)
        ctx.Ca, ctx.Cb = Ca, Cb
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        dy = dy.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty(dy.shape[:-1] + (ctx.Ca,), device=dy.device, dtype=dy.dtype)
            copy_channels(dy, 0, da, 0, ctx.Ca)
        if ctx.needs_input_grad[1]:
            db = torch.empty(dy.shape[:-1] + (ctx.Cb,), device=dy.device, dtype=dy.dtype)
            copy_channels(dy, ctx.Ca, db, 0, ctx.Cb)
        return da, db


def cat_channels(a, b):
    """torch.cat([a, b], dim=1) of the reference (NCHW) == last-dim concat in NHWC."""
    if TORCH_OPS_BOUNDARY:
        return torch.cat([a, b], dim=-1)
    return _CatFn.apply(a, b)


# ======================================================================================
# attention core (QKVAttentionLegacy)
# ======================================================================================
def _gemm_geom(M, N, K):
    return dict(B=1, H=1, W=M, Cin=K, Cout=N, R=1, S=1, pad=0, stride=1, Ho=1, Wo=M)


def _transpose_heads(qkv, coff, nh, ch):
    B, T, C3 = qkv.shape
    out = torch.empty((B * nh, ch, T), device=qkv.device, dtype=qkv.dtype)
    check(_lib.lib().jg_transpose_heads(_dt(qkv), qkv.data_ptr(), C3, coff, 3 * ch, out.data_ptr(), B, T, nh, ch, _st()),
          "jg_transpose_heads")
    return out


def attn_core_fwd(qkv, nh):
    """qkv [B,T,3C] in the LEGACY head layout (channel = h*3ch + {q: 0..ch, k: ch..2ch, v: 2ch..3ch},
    unet_generator_attn.py:340) -> (a [B,T,C] with channel = h*ch + c, P [B*nh,T,T] softmax probabilities)."""
    _require_cuda(qkv)
    L = _lib.lib()
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    ch = Cc // nh
    BH = B * nh
    dev, dt = qkv.device, _dt(qkv)
    if FLASH_ATTENTION and ch == 32 and T % 128 == 0:
        # fused kernel (attention.hip): the T x T logits stay on chip; aux = per-query logsumexp
        a = torch.empty((B, T, Cc), device=dev, dtype=qkv.dtype)
        lse = torch.empty((BH, T), device=dev, dtype=torch.float32)
        check(L.jg_attention_fwd(dt, qkv.data_ptr(), a.data_ptr(), lse.data_ptr(), B, T, nh, ch, _st()), "jg_attention_fwd")
        return a, lse
    scale2 = 1.0 / math.sqrt(ch)  # (ch^-1/4)^2
    S = torch.empty((BH, T, T), device=dev, dtype=torch.float32)
    conv_nt(qkv, qkv, S, **_gemm_geom(T, T, ch), ldx=C3, ldw=C3, ldy=T, alpha=scale2, out_f32=True, nbatch=BH, nh=nh,
            sx=(T * C3, 3 * ch), sw=(T * C3, 3 * ch), sy=(nh * T * T, T * T), w_off=ch)
    P = torch.empty((BH, T, T), device=dev, dtype=qkv.dtype)
    check(L.jg_softmax_fwd(dt, S.data_ptr(), P.data_ptr(), BH * T, T, _st()), "jg_softmax_fwd")
    del S
    Vt = _transpose_heads(qkv, 2 * ch, nh, ch)
    a = torch.empty((B, T, Cc), device=dev, dtype=qkv.dtype)
    conv_nt(P, Vt, a, **_gemm_geom(T, ch, T), ldx=T, ldw=T, ldy=Cc, nbatch=BH, nh=nh, sx=(nh * T * T, T * T),
            sw=(nh * ch * T, ch * T), sy=(T * Cc, ch))
    return a, P


def attn_core_bwd(qkv, P, da, nh, a=None):
    """Gradient of attn_core_fwd with respect to qkv.  `P` is the second result of attn_core_fwd (the
    probabilities [B*nh,T,T], or the logsumexp [B*nh,T] of the fused kernel, which also needs the output `a`)."""
    L = _lib.lib()
    da = da.contiguous()
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    ch = Cc // nh
    BH = B * nh
    dev, dt = qkv.device, _dt(qkv)
    if P.dim() == 2:
        if a is None:
            raise RuntimeError("the fused attention backward needs the forward output")
        dqkv = torch.empty_like(qkv)
        dq_rows = torch.empty((BH, T), device=dev, dtype=torch.float32)
        check(L.jg_attention_bwd(dt, qkv.data_ptr(), a.data_ptr(), P.data_ptr(), da.data_ptr(), dqkv.data_ptr(),
                                 dq_rows.data_ptr(), B, T, nh, ch, _st()), "jg_attention_bwd")
        return dqkv
    scale2 = 1.0 / math.sqrt(ch)
    # dP = dA V^T
    dP = torch.empty((BH, T, T), device=dev, dtype=torch.float32)
    conv_nt(da, qkv, dP, **_gemm_geom(T, T, ch), ldx=Cc, ldw=C3, ldy=T, out_f32=True, nbatch=BH, nh=nh,
            sx=(T * Cc, ch), sw=(T * C3, 3 * ch), sy=(nh * T * T, T * T), w_off=2 * ch)
    dS = torch.empty((BH, T, T), device=dev, dtype=qkv.dtype)
    check(L.jg_softmax_bwd(dt, P.data_ptr(), dP.data_ptr(), dS.data_ptr(), BH * T, T, scale2, _st()), "jg_softmax_bwd")
    del dP
    dqkv = torch.empty_like(qkv)
    # dQ = dS K      (NT with K^T as the "weight" operand)
    Kt = _transpose_heads(qkv, ch, nh, ch)
    conv_nt(dS, Kt, dqkv, **_gemm_geom(T, ch, T), ldx=T, ldw=T, ldy=C3, nbatch=BH, nh=nh, sx=(nh * T * T, T * T),
            sw=(nh * ch * T, ch * T), sy=(T * C3, 3 * ch))
    # dK = dS^T Q ; dV = P^T dA     (TN: reduction index t is the row index of both operands)
    geom = dict(B=1, H=1, W=T, Cin=ch, Cout=T, R=1, S=1, pad=0, stride=1, Ho=1, Wo=T)
    wgrad_tn(dS, qkv, dqkv, **geom, lddy=T, ldx=C3, lddw=C3, nbatch=BH, nh=nh, sdy=(nh * T * T, T * T),
             sx=(T * C3, 3 * ch), sdw=(T * C3, 3 * ch), out_mode=JG_OUT_STORE_T, dw_off=ch)
    wgrad_tn(P, da, dqkv, **geom, lddy=T, ldx=Cc, lddw=C3, nbatch=BH, nh=nh, sdy=(nh * T * T, T * T),
             sx=(T * Cc, ch), sdw=(T * C3, 3 * ch), out_mode=JG_OUT_STORE_T, dw_off=2 * ch)
    return dqkv


class _AttnCoreFn(JGFunction):
    @staticmethod
    def forward(ctx, qkv, nh):
        a, P = attn_core_fwd(qkv, nh)
        ctx.save_for_backward(qkv, P, a)
        ctx.nh = nh
        return a

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, da):
        qkv, P, a = ctx.saved_tensors
        return attn_core_bwd(qkv, P, da, ctx.nh, a), None


def attention_core(qkv, n_heads):
    return torch.ops.jg355.attention_core(qkv, n_heads)[0]


# ======================================================================================
# small fp32 linear (embedding path)
# ======================================================================================
class _LinearFn(JGFunction):
    """inputs: x, W, b, act, dW, db, *track -- dW/db are the arena gradient views the kernel accumulates
    into; `track` are the nn.Parameters behind W/b, passed only so that autograd records the node."""

    @staticmethod
    def forward(ctx, x, W, b, act, dW, db, *track):
        _require_cuda(x, W)
        x = x.contiguous()
        Bn, K = x.shape
        N = W.shape[0]
        y = torch.empty((Bn, N), device=x.device, dtype=torch.float32)
        check(_lib.lib().jg_linear_fwd(x.data_ptr(), W.data_ptr(), _p(b), y.data_ptr(), Bn, K, N, act, _st()),
              "jg_linear_fwd")
        ctx.save_for_backward(x, W)
        ctx.act, ctx.dW, ctx.db = act, dW, db
        ctx.want_param = any(t is not None and t.requires_grad for t in track)
        ctx.ntrack = len(track)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        Bn, K = x.shape
        N = W.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = db = None
        if ctx.want_param:
            dW, db = ctx.dW, ctx.db
            if dW is None:
                raise RuntimeError("linear weight has no arena-backed .grad")
        if dx is not None or dW is not None:
            check(_lib.lib().jg_linear_bwd(x.data_ptr(), W.data_ptr(), dy.data_ptr(), _p(dx), _p(dW), _p(db), Bn, K, N,
                                           ctx.act, _st()), "jg_linear_bwd")
        return (dx, None, None, None, None, None) + (None,) * ctx.ntrack


def linear(x, weight, bias=None, act=JG_ACT_NONE):
    """y = act(x) @ weight.T + bias in fp32 (nn.Linear preceded by an optional SiLU)."""
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.linear_act(x, weight, bias, act)
    track = (weight,) if bias is None else (weight, bias)
    return _LinearFn.apply(x, weight, bias, act, weight.grad, None if bias is None else bias.grad, *track)


def linear_stacked(x, W, b, dW, db, act, track):
    """Several nn.Linear layers that share their input, run as one: W/b/dW/db are stacked arena views."""
    return _LinearFn.apply(x, W, b, act, dW, db, *track)


def gamma_embedding(gammas, dim, max_period=10000.0):
    """models/modules/diffusion_utils.py:8-42 for gammas [B,1] (or [B])."""
    if TORCH_OPS_BOUNDARY and not _IN_OP:
        return torch.ops.jg355.gamma_embedding(gammas, dim, float(max_period))
    g = gammas.reshape(-1).contiguous().float()
    emb = torch.empty((g.shape[0], dim), device=g.device, dtype=torch.float32)
    check(_lib.lib().jg_gamma_embedding(g.data_ptr(), emb.data_ptr(), g.shape[0], dim, float(max_period), _st()),
          "jg_gamma_embedding")
    return emb


# ======================================================================================
# CUT patch features and contrastive / adversarial losses (fp32)
# ======================================================================================
def sgemm(A, B, C, M, N, K, sa, sb, sc, nbatch=1, bstr=(0, 0, 0), bias=None, E=None, alpha=1.0, beta=0.0,
          act_a=JG_ACT_NONE, act_b=JG_ACT_NONE, act_e=JG_ACT_NONE):
    """C[z][m][n] = alpha sum_k actA(A[z][m][k]) actB(B[z][n][k]) (+bias, *act'(E), +beta C); sa=(sam,sak), sb=(sbn,sbk), sc=(scm,scn)."""
    check(_lib.lib().jg_sgemm(A.data_ptr(), B.data_ptr(), C.data_ptr(), _p(bias), _p(E), M, N, K, sa[0], sa[1], sb[0], sb[1],
                              sc[0], sc[1], nbatch, bstr[0], bstr[1], bstr[2], alpha, beta, act_a, act_b, act_e, _st()), "jg_sgemm")
    return C


class _GatherPatchesFn(JGFunction):
    @staticmethod
    def forward(ctx, feat, ids, C, per):
        _require_cuda(feat, ids)
        B, H, W, ld = feat.shape
        G, P = (ids.shape[0], ids.shape[1]) if ids.dim() == 2 else (1, ids.numel())
        per = per if G > 1 else B
        out = torch.empty((B * P, C), device=feat.device, dtype=torch.float32)
        check(_lib.lib().jg_gather_rows_grouped(_dt(feat), feat.data_ptr(), ld, ids.data_ptr(), out.data_ptr(), B, H * W, C, P, G, per, _st()),
              "jg_gather_rows_grouped")
        ctx.save_for_backward(ids)
        ctx.shape, ctx.dtype, ctx.C, ctx.geo = feat.shape, feat.dtype, C, (G, P, per)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        B, H, W, ld = ctx.shape
        G, P, per = ctx.geo
        dfeat = torch.zeros(ctx.shape, device=dout.device, dtype=ctx.dtype)
        dout = dout.contiguous()
        check(_lib.lib().jg_scatter_rows_grouped(_DT[ctx.dtype], dfeat.data_ptr(), ld, ids.data_ptr(), dout.data_ptr(), B, H * W, ctx.C, P, G, per,
                                                 _st()), "jg_scatter_rows_grouped")
        return dfeat, None, None, None


def gather_patches(feat, ids, C, per=0):
    """feat [B,H,W,ld] 16-bit NHWC -> [B*P, C] fp32 rows at the flattened positions `ids` (shared by the batch):
    `feat.permute(0,2,3,1).flatten(1,2)[:, patch_id, :].flatten(0,1)` of cut_networks.py:45-57.  `ids` [G, P] with `per` > 0: G id sets,
    image b uses set (b // per) % G (one launch for the concatenated batches of several PatchSampleF calls)."""
    if ids.dim() == 2 and ids.shape[0] > 1:
        if per < 1 or TORCH_OPS_BOUNDARY:
            raise ValueError("grouped gather_patches needs per >= 1 and the ctypes path")
        return _GatherPatchesFn.apply(feat, ids.contiguous().long(), C, int(per))
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.gather_patches(feat, ids.contiguous().long(), C)
    return _GatherPatchesFn.apply(feat, ids.reshape(-1).contiguous().long(), C, 0)


class _L2NormFn(JGFunction):
    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        R, D = x.shape
        y = torch.empty_like(x)
        nrm = torch.empty(R, device=x.device, dtype=torch.float32)
        check(_lib.lib().jg_l2norm_fwd(x.data_ptr(), y.data_ptr(), nrm.data_ptr(), R, D, eps, _st()), "jg_l2norm_fwd")
        ctx.save_for_backward(y, nrm)
        ctx.eps = eps
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(_lib.lib().jg_l2norm_bwd(y.data_ptr(), nrm.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.shape[0], y.shape[1], ctx.eps,
                                       _st()), "jg_l2norm_bwd")
        return dx, None


def l2_normalize(x, eps=1e-7):
    """torch.nn.functional.normalize(x, eps=eps) for fp32 [R, D] (cut_networks.py:66)."""
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.l2_normalize(x, eps)[0]
    return _L2NormFn.apply(x, eps)


SINKHORN_ITERS = 50  # monce.py:24


class _PatchNCEFn(JGFunction):
    """Per-patch PatchNCE / MoNCE loss.  q, k: [nimg*P, D] fp32, L2-normalised.  k is detached in the positive logit and in the
    optimal-transport weights but NOT in the negative logits (base_NCE.py:52-66, monce.py:21-22), reproduced here."""

    @staticmethod
    def forward(ctx, q, k, nimg, T, pm1, monce):
        _require_cuda(q, k)
        q, k = q.contiguous(), k.contiguous()
        R, D = q.shape
        P = R // nimg
        L = _lib.lib()
        S = torch.empty((nimg, P, P), device=q.device, dtype=torch.float32)
        sgemm(q, k, S, P, P, D, (D, 1), (D, 1), (P, 1), nimg, (P * D, P * D, P * P))
        loss = torch.empty(R, device=q.device, dtype=torch.float32)
        K = uh = vh = None
        if monce:
            K = torch.empty_like(S)
            uh = torch.empty((nimg, SINKHORN_ITERS, P), device=q.device, dtype=torch.float32)
            vh = torch.empty((nimg, SINKHORN_ITERS + 1, P), device=q.device, dtype=torch.float32)
            check(L.jg_nce_sinkhorn_fwd(S.data_ptr(), K.data_ptr(), uh.data_ptr(), vh.data_ptr(), nimg, P, SINKHORN_ITERS, 1.0, _st()),
                  "jg_nce_sinkhorn_fwd")
        u = uh[:, -1] if monce else None
        v = vh[:, -1] if monce else None
        check(L.jg_nce_ce(S.data_ptr(), _p(u), SINKHORN_ITERS * P, _p(v), (SINKHORN_ITERS + 1) * P, loss.data_ptr(), None, None,
                          nimg, P, T, pm1, None, None, 1.0, _st()), "jg_nce_ce")
        ctx.save_for_backward(q, k, S, K, uh, vh)
        ctx.cfg = (nimg, P, D, T, pm1, monce)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dloss):
        q, k, S, K, uh, vh = ctx.saved_tensors
        nimg, P, D, T, pm1, monce = ctx.cfg
        L = _lib.lib()
        dloss = dloss.contiguous().float()
        dS = torch.empty_like(S)
        gW = torch.empty_like(S) if monce else None
        gpos = torch.empty(nimg * P, device=q.device, dtype=torch.float32)
        scratch = torch.empty_like(q)
        u = uh[:, -1] if monce else None
        v = vh[:, -1] if monce else None
        check(L.jg_nce_ce(S.data_ptr(), _p(u), SINKHORN_ITERS * P, _p(v), (SINKHORN_ITERS + 1) * P, scratch.data_ptr(), dS.data_ptr(),
                          _p(gW), nimg, P, T, pm1, dloss.data_ptr(), gpos.data_ptr(), 1.0, _st()), "jg_nce_ce")
        dk = None
        if ctx.needs_input_grad[1]:
            # dk_j = sum_i dS_ij q_i over the negatives only (diagonal of dS is zero, the positive pair is detached)
            dk = torch.empty_like(k)
            sgemm(dS, q, dk, P, D, P, (1, P), (1, D), (D, 1), nimg, (P * P, P * D, P * D))
        if monce:
            dsh = torch.empty_like(uh)
            drh = torch.empty_like(uh)
            check(L.jg_nce_sinkhorn_bwd(K.data_ptr(), uh.data_ptr(), vh.data_ptr(), gW.data_ptr(), dsh.data_ptr(), drh.data_ptr(),
                                        dS.data_ptr(), nimg, P, SINKHORN_ITERS, _st()), "jg_nce_sinkhorn_bwd")
        dq = torch.empty_like(q)
        sgemm(dS, k, dq, P, D, P, (P, 1), (1, D), (D, 1), nimg, (P * P, P * D, P * D))
        check(L.jg_row_axpy(dq.data_ptr(), gpos.data_ptr(), k.data_ptr(), nimg * P, D, _st()), "jg_row_axpy")
        return dq, dk, None, None, None, None


def patch_nce_loss(q, k, nimg, T, num_patches, monce=False):
    """BaseNCELoss.forward / MoNCELoss (base_NCE.py:17-50, monce.py:16-33): per-patch loss [nimg*P]."""
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.patch_nce(q, k, nimg, float(T), float(num_patches - 1), bool(monce))[0]
    return _PatchNCEFn.apply(q, k, nimg, float(T), float(num_patches - 1), bool(monce))


GAN_MODES = {"lsgan": 0, "vanilla": 1, "wgangp": 2}


class _LSGANLossFn(JGFunction):
    @staticmethod
    def forward(ctx, pred, target, scale, mode=0):
        _require_cuda(pred)
        pred = pred.contiguous()
        cpad = pred.shape[-1]
        npix = pred.numel() // cpad
        loss = torch.zeros((), device=pred.device, dtype=torch.float32)
        dpred = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        check(_lib.lib().jg_gan_loss(_dt(pred), mode, pred.data_ptr(), target, loss.data_ptr(), _p(dpred), npix, cpad, scale, 1.0, _st()),
              "jg_gan_loss")
        ctx.dpred = dpred
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        dpred, ctx.dpred = ctx.dpred, None
        return axpby(dpred, 1.0, None, 0.0, alpha_dev=g.reshape(1).float(), out=dpred), None, None, None


def lsgan_loss(pred, target, scale=1.0):
    """GANLoss('lsgan') (loss.py:69-71): scale * mean((pred[..., 0] - target)^2) on an NHWC logit map whose channel 0 is valid."""
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.gan_loss(pred, 0, float(target), float(scale))[0]
    return _LSGANLossFn.apply(pred, float(target), float(scale), 0)


def gan_loss(pred, mode, target, scale=1.0):
    """GANLoss(mode) for mode in lsgan / vanilla / wgangp (loss.py:59-76) on an NHWC logit map whose channel 0 is valid; `target` is the label
    (real_label / fake_label)."""
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.gan_loss(pred, GAN_MODES[mode], float(target), float(scale))[0]
    return _LSGANLossFn.apply(pred, float(target), float(scale), GAN_MODES[mode])


# ======================================================================================
# DDPM glue + layout converters
# ======================================================================================
_IN_OP = False      # set while a torch.ops.jg355 implementation calls back into the plain helpers of this module


def ddpm_prepare(y0, ycond, noise, mask, gammas, act_dtype, cpad=8):
    if TORCH_OPS_BOUNDARY and not _IN_OP:
        return torch.ops.jg355.ddpm_prepare(y0, ycond, noise, mask, gammas, act_dtype == torch.float16, cpad)
    _require_cuda(y0, ycond, noise)
    B, Cc, H, W = y0.shape
    xin = torch.empty((B, H, W, cpad), device=y0.device, dtype=act_dtype)
    m = None
    if mask is not None:
        m = mask.contiguous()
        if m.dtype != torch.int64:
            m = m.long()
    check(_lib.lib().jg_ddpm_prepare(_DT[act_dtype], y0.contiguous().data_ptr(), ycond.contiguous().data_ptr(),
                                     noise.contiguous().data_ptr(), _p(m), gammas.contiguous().data_ptr(), xin.data_ptr(), B,
                                     Cc, H, W, cpad, _st()), "jg_ddpm_prepare")
    return xin


class _MSELossFn(JGFunction):
    @staticmethod
    def forward(ctx, nh, noise, mask, w, lam, grad_scale, Cc):
        B, H, W, cpad = nh.shape
        loss = torch.zeros((), device=nh.device, dtype=torch.float32)
        dnh = torch.empty_like(nh)
        check(_lib.lib().jg_ddpm_mse_loss(_dt(nh), noise.data_ptr(), nh.data_ptr(), _p(mask), _p(w), loss.data_ptr(),
                                          dnh.data_ptr(), B, Cc, H, W, cpad, lam, grad_scale, _st()), "jg_ddpm_mse_loss")
        ctx.save_for_backward(dnh)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        (dnh,) = ctx.saved_tensors
        gout = gout.contiguous().float()
        return axpby(dnh, 1.0, alpha_dev=gout), None, None, None, None, None, None


def ddpm_mse_loss(noise_hat_nhwc, noise, mask, w=None, lam=1.0, grad_scale=1.0):
    """lambda * MSE(w m noise, w m noise_hat) with its gradient produced in the same pass."""
    m = None
    if mask is not None:
        m = mask.contiguous()
        if m.dtype != torch.int64:
            m = m.long()
    wv = None if w is None else w.reshape(-1).contiguous().float()
    if TORCH_OPS_BOUNDARY:
        return torch.ops.jg355.ddpm_mse_loss(noise_hat_nhwc, noise.contiguous(), m, wv, float(lam), float(grad_scale), noise.shape[1])[0]
    return _MSELossFn.apply(noise_hat_nhwc, noise.contiguous(), m, wv, float(lam), float(grad_scale), noise.shape[1])


class _MultiScaleLossFn(JGFunction):
    @staticmethod
    def forward(ctx, nh, noise, mask, w, lam, grad_scale, Cc, nlevels, l1, multiscale):
        B, H, W, cpad = nh.shape
        losses = torch.zeros(nlevels, device=nh.device, dtype=torch.float32)
        dnh = torch.empty_like(nh)
        nws = sum(B * Cc * (H >> l) * (W >> l) for l in range(1, nlevels))
        ws = torch.empty(max(nws, 1), device=nh.device, dtype=torch.float32)
        check(_lib.lib().jg_ddpm_multiscale_loss(_dt(nh), noise.data_ptr(), nh.data_ptr(), _p(mask), _p(w), losses.data_ptr(),
                                                 dnh.data_ptr(), ws.data_ptr(), B, Cc, H, W, cpad, nlevels, int(l1), int(multiscale), lam,
                                                 grad_scale, _st()), "jg_ddpm_multiscale_loss")
        ctx.save_for_backward(dnh)
        ctx.mark_non_differentiable(losses)
        return losses.sum(), losses

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout, _g_levels):
        (dnh,) = ctx.saved_tensors
        return (axpby(dnh, 1.0, alpha_dev=gout.contiguous().float().reshape(1)),) + (None,) * 9


def ddpm_loss(noise_hat_nhwc, noise, mask, w=None, lam=1.0, grad_scale=1.0, lossname="MSE", min_res=32):
    """alg_palette_loss in {MSE, L1, multiscale_MSE, multiscale_L1} (palette_model.py:231-256,597-618): returns
    (lambda * total, {resolution: lambda * term}) -- the per-resolution terms only for the multiscale variants."""
    if lossname == "MSE":
        return ddpm_mse_loss(noise_hat_nhwc, noise, mask, w, lam, grad_scale), {}
    m = None
    if mask is not None:
        m = mask.contiguous()
        if m.dtype != torch.int64:
            m = m.long()
    wv = None if w is None else w.reshape(-1).contiguous().float()
    S = noise_hat_nhwc.shape[1]
    multiscale = lossname.startswith("multiscale")
    if min_res != 32:
        raise NotImplementedError("MultiScaleDiffusionLoss.min_res is 32 in the reference")
    nlevels = 1
    if multiscale:
        import math

        nlevels = max(1, math.floor(math.log2(S)) - 5 + 1)        # resolutions 32 .. S (loss.py:406-409, palette_model.py:232-241)
    tot, levels = _MultiScaleLossFn.apply(noise_hat_nhwc, noise.contiguous(), m, wv, float(lam), float(grad_scale), noise.shape[1],
                                          nlevels, lossname.endswith("L1"), multiscale)
    return tot, ({str(S >> l): levels[l] for l in range(nlevels)} if multiscale else {})


# ======================================================================================
# consistency-model glue (cm_generator.py / cm_model.py of the reference)
# ======================================================================================
class _NoiseLevelEmbFn(JGFunction):
    @staticmethod
    def forward(ctx, sigmas, W, dW):
        s = sigmas.reshape(-1).contiguous().float()
        emb = torch.empty((s.shape[0], 2 * W.shape[0]), device=s.device, dtype=torch.float32)
        check(_lib.lib().jg_noise_level_embedding(s.data_ptr(), W.data_ptr(), emb.data_ptr(), s.shape[0], W.shape[0], _st()),
              "jg_noise_level_embedding")
        ctx.save_for_backward(s, W)
        ctx.dW = dW
        return emb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, demb):
        s, W = ctx.saved_tensors
        if ctx.dW is not None:
            demb = demb.contiguous().float()
            check(_lib.lib().jg_noise_level_embedding_bwd(s.data_ptr(), W.data_ptr(), demb.data_ptr(), ctx.dW.data_ptr(),
                                                          s.shape[0], W.shape[0], _st()), "jg_noise_level_embedding_bwd")
        return None, None, None


def noise_level_embedding(sigmas, W):
    """[sin | cos](sigma * W * 2 pi) (NoiseLevelEmbedding.forward, cm_generator.py:276-280); the gradient with respect
    to W is accumulated into W.grad (arena view) by the backward kernel."""
    _require_cuda(sigmas, W)
    return _NoiseLevelEmbFn.apply(sigmas, W, W.grad if W.requires_grad else None)


def cm_noisy(x, noise, sigma, mask, cond, act_dtype, cpad=8):
    """(noisy fp32 NCHW, UNet input 16-bit NHWC [cond | noisy | 0]) -- cm_generator.py:452-460,377-381."""
    _require_cuda(x, noise, sigma)
    B, Cc, H, W = x.shape
    Ccond = 0 if cond is None else cond.shape[1]
    m = None
    if mask is not None:
        m = mask.contiguous()
        if m.dtype != torch.int64:
            m = m.long()
    out = torch.empty_like(x, dtype=torch.float32)
    xin = torch.empty((B, H, W, cpad), device=x.device, dtype=act_dtype)
    check(_lib.lib().jg_cm_noisy(_DT[act_dtype], x.contiguous().float().data_ptr(), noise.contiguous().float().data_ptr(),
                                 sigma.contiguous().float().data_ptr(), _p(m), None if cond is None else cond.contiguous().float().data_ptr(),
                                 out.data_ptr(), xin.data_ptr(), B, Cc, Ccond, H, W, cpad, _st()), "jg_cm_noisy")
    return out, xin


def cm_combine(noisy, F_nhwc, cskip, cout):
    """c_skip * noisy + c_out * F as fp32 NCHW (cm_forward, cm_generator.py:383-385); no autograd (outputs / visuals)."""
    B, Cc, H, W = noisy.shape
    out = torch.empty_like(noisy)
    check(_lib.lib().jg_cm_combine(_dt(F_nhwc), noisy.data_ptr(), F_nhwc.data_ptr(), cskip.contiguous().data_ptr(),
                                   cout.contiguous().data_ptr(), out.data_ptr(), B, Cc, H, W, F_nhwc.shape[-1], _st()), "jg_cm_combine")
    return out


class _CMLossFn(JGFunction):
    @staticmethod
    def forward(ctx, Fn, Fc, noisy_n, noisy_c, cs_n, co_n, cs_c, co_c, mask, w, chub, lam, grad_scale):
        B, Cc, H, W = noisy_n.shape
        loss = torch.zeros((), device=Fn.device, dtype=torch.float32)
        dFn = torch.empty_like(Fn)
        check(_lib.lib().jg_cm_loss(_dt(Fn), Fn.data_ptr(), Fc.data_ptr(), noisy_n.data_ptr(), noisy_c.data_ptr(), cs_n.data_ptr(),
                                    co_n.data_ptr(), cs_c.data_ptr(), co_c.data_ptr(), _p(mask), w.data_ptr(), loss.data_ptr(),
                                    dFn.data_ptr(), B, Cc, H, W, Fn.shape[-1], chub, lam, grad_scale, _st()), "jg_cm_loss")
        ctx.save_for_backward(dFn)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        (dFn,) = ctx.saved_tensors
        return (axpby(dFn, 1.0, alpha_dev=gout.contiguous().float()),) + (None,) * 12


def cm_loss(F_next, F_cur, noisy_next, noisy_cur, cs_n, co_n, cs_c, co_c, mask, loss_weights, lam=1.0, grad_scale=1.0):
    """compute_cm_loss (cm_model.py:353-375, no perceptual terms) on the two UNet outputs (NHWC 16-bit); gradient with
    respect to F_next produced in the same pass."""
    B, Cc, H, W = noisy_next.shape
    m = None
    if mask is not None:
        m = mask.contiguous()
        if m.dtype != torch.int64:
            m = m.long()
    chub = 0.00054 * math.sqrt(Cc * H * W)
    f = lambda t: t.reshape(-1).contiguous().float()
    return _CMLossFn.apply(F_next, F_cur.detach(), noisy_next, noisy_cur, f(cs_n), f(co_n), f(cs_c), f(co_c), m, f(loss_weights),
                           chub, float(lam), float(grad_scale))


class _ToNCHWFn(JGFunction):
    @staticmethod
    def forward(ctx, x, Cc):
        B, H, W, cpad = x.shape
        y = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
        check(_lib.lib().jg_nhwc_to_nchw_f32(_dt(x), x.data_ptr(), y.data_ptr(), B, Cc, H, W, cpad, _st()), "nhwc_to_nchw")
        ctx.cpad, ctx.dtype = cpad, x.dtype
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        return to_nhwc(dy.contiguous().float(), ctx.dtype, ctx.cpad), None


def to_nchw_f32(x, Cc):
    return _ToNCHWFn.apply(x, Cc)


def to_nhwc(x, act_dtype, cpad=None):
    """NCHW fp32 -> NHWC 16-bit with the channel count padded to `cpad` (multiple of 8). No autograd."""
    B, Cc, H, W = x.shape
    cpad = cpad or (Cc + 7) // 8 * 8
    y = torch.empty((B, H, W, cpad), device=x.device, dtype=act_dtype)
    check(_lib.lib().jg_nchw_f32_to_nhwc(_DT[act_dtype], x.contiguous().data_ptr(), y.data_ptr(), B, Cc, H, W, cpad, _st()),
          "nchw_to_nhwc")
    return y


# ======================================================================================
# torch.ops.jg355.* -- the op surface north_star names: every op is a torch.library custom op with a fake (meta) kernel and an
# autograd formula built from the C-ABI kernels, so the ops trace (torch.compile / FakeTensor), compose with autograd and pass
# torch.library.opcheck.  The ops are FUNCTIONAL (gradients are returned).  The module graph uses them where a functional op costs
# nothing (attention core, pooling, upsampling); convolutions and normalisations of the training path keep their arena-accumulating
# autograd nodes (`_Conv2dFn`, `_GroupNormFn`, modules/unet_exec.py), whose weight / affine gradients are atomically added IN PLACE
# into the flat gradient arena -- returning them as tensors would add a full-arena copy to every step.
# ======================================================================================
def _conv_out_hw(H, W, R, S, pad, stride):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


@torch.library.custom_op("jg355::conv2d_nt", mutates_args=())
def _op_conv2d_nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], res: Optional[torch.Tensor], pad: int, stride: int,
                  alpha: float, res_scale: float) -> torch.Tensor:
    """x [B,H,W,Cin] 16-bit NHWC, w [Cout,R,S,Cin] 16-bit (or the fp32 master weight: rounded to x's type inside, gradient returned in
    fp32), bias fp32 [Cout], res like y: y = alpha conv(x, w) + bias + res_scale res"""
    B, H, W, Cin = x.shape
    Cout, R, S, _ = w.shape
    Ho, Wo = _conv_out_hw(H, W, R, S, pad, stride)
    y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=x.dtype)
    x, w = x.contiguous(), w.to(x.dtype).contiguous()       # an fp32 master weight is rounded here, as the arena's working copy is
    conv_nt(x, w, y, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=S, pad=pad, stride=stride, Ho=Ho, Wo=Wo, ldx=Cin,
            ldw=R * S * Cin, ldy=Cout, bias=bias, res=None if res is None else res.contiguous(), ldres=Cout, alpha=alpha, res_scale=res_scale)
    return y


@_op_conv2d_nt.register_fake
def _(x, w, bias, res, pad, stride, alpha, res_scale):
    B, H, W, _ = x.shape
    Cout, R, S, _ = w.shape
    Ho, Wo = _conv_out_hw(H, W, R, S, pad, stride)
    return x.new_empty((B, Ho, Wo, Cout))


@torch.library.custom_op("jg355::conv2d_wgrad", mutates_args=())
def _op_conv2d_wgrad(dy: torch.Tensor, x: torch.Tensor, R: int, S: int, pad: int, stride: int, alpha: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dW fp32 [Cout,R,S,Cin], dbias fp32 [Cout]) of y = alpha conv(x, W) + bias for the output gradient dy"""
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    dw = torch.zeros((Cout, R, S, Cin), device=x.device, dtype=torch.float32)
    db = torch.zeros((Cout,), device=x.device, dtype=torch.float32)
    ktot = R * S * Cin
    splitk = _wgrad_splitk(((Cout + 127) // 128) * ((ktot + 127) // 128), B * Ho * Wo)
    wgrad_tn(dy.contiguous(), x.contiguous(), dw, B=B, H=H, W=W, Cin=Cin, Cout=Cout, R=R, S=S, pad=pad, stride=stride, Ho=Ho, Wo=Wo,
             lddy=Cout, ldx=Cin, lddw=ktot, dbias=db, splitk=splitk, alpha=alpha)
    return dw, db


@_op_conv2d_wgrad.register_fake
def _(dy, x, R, S, pad, stride, alpha):
    return x.new_empty((dy.shape[-1], R, S, x.shape[-1]), dtype=torch.float32), x.new_empty((dy.shape[-1],), dtype=torch.float32)


def _conv2d_nt_setup(ctx, inputs, output):
    x, w, bias, res, pad, stride, alpha, res_scale = inputs
    ctx.save_for_backward(x, w)
    ctx.geo = (pad, stride, alpha, res_scale, bias is not None, res is not None)


def _conv2d_nt_backward(ctx, dy):
    x, w = ctx.saved_tensors
    pad, stride, alpha, res_scale, has_bias, has_res = ctx.geo
    Cout, R, S, Cin = w.shape
    dy = dy.contiguous()
    dx = dw = db = dres = None
    if ctx.needs_input_grad[0]:          # input gradient = the same kernel on the flipped / transposed weights
        wT = w.to(dy.dtype).permute(3, 1, 2, 0).flip(1, 2).contiguous()
        dyd = dy
        if stride != 1:                  # stride s: over the zero-dilated dy (length H + 2p - k + 1 per axis), as ops.conv2d_dgrad
            H, W_ = x.shape[1], x.shape[2]
            dyd = torch.ops.jg355.dilate2d(dy, H + 2 * pad - R + 1, W_ + 2 * pad - S + 1, stride)
        dx = torch.ops.jg355.conv2d_nt(dyd, wT, None, None, R - 1 - pad, 1, alpha, 0.0)
    if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
        dwf, dbf = torch.ops.jg355.conv2d_wgrad(dy, x, R, S, pad, stride, alpha)
        dw = dwf.to(w.dtype) if ctx.needs_input_grad[1] else None
        db = dbf if (has_bias and ctx.needs_input_grad[2]) else None
    if has_res and ctx.needs_input_grad[3]:
        dres = dy if res_scale == 1.0 else (dy.float() * res_scale).to(dy.dtype)
    return dx, dw, db, dres, None, None, None, None


_op_conv2d_nt.register_autograd(_conv2d_nt_backward, setup_context=_conv2d_nt_setup)


@torch.library.custom_op("jg355::group_norm_act", mutates_args=())
def _op_group_norm_act(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], film: Optional[torch.Tensor],
                       groups: int, act: int, eps: float) -> torch.Tensor:
    """act(GroupNorm_groups(x) gamma + beta [(1 + scale) + shift]); x NHWC 16-bit (any number of middle dims), film fp32 [B, 2C]"""
    return _GroupNormFn.forward(_NoCtx(), x.contiguous(), gamma, beta, None if film is None else film.contiguous(), groups, act, eps)


class _NoCtx:
    """stand-in autograd context for calling a Function's forward as a plain kernel sequence"""
    needs_input_grad = (False,) * 8

    def save_for_backward(self, *a):
        self.saved = a


@_op_group_norm_act.register_fake
def _(x, gamma, beta, film, groups, act, eps):
    return torch.empty_like(x)


@torch.library.custom_op("jg355::group_norm_act_bwd", mutates_args=())
def _op_group_norm_act_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor],
                           film: Optional[torch.Tensor], groups: int, act: int, eps: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(dx, dgamma [C], dbeta [C], dfilm [B, 2C]) -- the statistics are recomputed from x (two small kernels)"""
    L = _lib.lib()
    x, dy = x.contiguous(), dy.contiguous()
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    dev, st, dt = x.device, _st(), _dt(x)
    f32 = dict(device=dev, dtype=torch.float32)
    sums, ab, mr = torch.empty((B, C, 2), **f32), torch.empty((B, C, 2), **f32), torch.empty((B, groups, 2), **f32)
    film = None if film is None else film.contiguous()
    ldfilm = film.stride(0) if film is not None else 0
    check(L.jg_gn_stats(dt, x.data_ptr(), sums.data_ptr(), B, HW, C, st), "jg_gn_stats")
    check(L.jg_gn_coef(sums.data_ptr(), _p(gamma), _p(beta), _p(film), ldfilm, ab.data_ptr(), mr.data_ptr(), B, HW, C, groups, eps, st), "jg_gn_coef")
    red, pqr = torch.empty((B, C, 2), **f32), torch.empty((B, C, 3), **f32)
    dgamma, dbeta, dfilm = torch.zeros((C,), **f32), torch.zeros((C,), **f32), torch.zeros((B, 2 * C), **f32)
    check(L.jg_gn_bwd_reduce(dt, x.data_ptr(), dy.data_ptr(), ab.data_ptr(), red.data_ptr(), B, HW, C, act, st), "jg_gn_bwd_reduce")
    check(L.jg_gn_bwd_coef(red.data_ptr(), _p(gamma), _p(beta), _p(film), ldfilm, mr.data_ptr(), pqr.data_ptr(),
                           dgamma.data_ptr() if gamma is not None else None, dbeta.data_ptr() if beta is not None else None,
                           dfilm.data_ptr() if film is not None else None, 2 * C, B, HW, C, groups, st), "jg_gn_bwd_coef")
    dx = torch.empty_like(x)
    check(L.jg_gn_bwd_apply(dt, x.data_ptr(), dy.data_ptr(), ab.data_ptr(), pqr.data_ptr(), dx.data_ptr(), B, HW, C, act, st), "jg_gn_bwd_apply")
    return dx, dgamma, dbeta, dfilm


@_op_group_norm_act_bwd.register_fake
def _(x, dy, gamma, beta, film, groups, act, eps):
    B, C = x.shape[0], x.shape[-1]
    f = lambda *sh: x.new_empty(sh, dtype=torch.float32)
    return torch.empty_like(x), f(C), f(C), f(B, 2 * C)


def _gn_setup(ctx, inputs, output):
    x, gamma, beta, film, groups, act, eps = inputs
    ctx.save_for_backward(x, gamma, beta, film)
    ctx.cfg = (groups, act, eps)


def _gn_backward(ctx, dy):
    x, gamma, beta, film = ctx.saved_tensors
    groups, act, eps = ctx.cfg
    dx, dgamma, dbeta, dfilm = torch.ops.jg355.group_norm_act_bwd(x, dy, gamma, beta, film, groups, act, eps)
    return (dx if ctx.needs_input_grad[0] else None, dgamma if (gamma is not None and ctx.needs_input_grad[1]) else None,
            dbeta if (beta is not None and ctx.needs_input_grad[2]) else None,
            dfilm if (film is not None and ctx.needs_input_grad[3]) else None, None, None, None)


_op_group_norm_act.register_autograd(_gn_backward, setup_context=_gn_setup)


@torch.library.custom_op("jg355::attention_core", mutates_args=())
def _op_attention_core(qkv: torch.Tensor, heads: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """QKVAttentionLegacy core: qkv [B,T,3C] (legacy head layout) -> (a [B,T,C], aux) with aux = per-query logsumexp of the fused
    kernel (head dim 32, T % 128 == 0) or the softmax probabilities of the unfused path (what the backward needs)"""
    a, P = attn_core_fwd(qkv.contiguous(), heads)
    return a, P


@_op_attention_core.register_fake
def _(qkv, heads):
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    a = qkv.new_empty((B, T, Cc))
    if FLASH_ATTENTION and Cc // heads == 32 and T % 128 == 0:
        return a, qkv.new_empty((B * heads, T), dtype=torch.float32)
    return a, qkv.new_empty((B * heads, T, T))


@torch.library.custom_op("jg355::attention_core_bwd", mutates_args=())
def _op_attention_core_bwd(qkv: torch.Tensor, aux: torch.Tensor, a: torch.Tensor, da: torch.Tensor, heads: int) -> torch.Tensor:
    return attn_core_bwd(qkv.contiguous(), aux, da, heads, a)


@_op_attention_core_bwd.register_fake
def _(qkv, aux, a, da, heads):
    return torch.empty_like(qkv)


def _attn_setup(ctx, inputs, output):
    qkv, heads = inputs
    a, aux = output
    ctx.save_for_backward(qkv, aux, a)
    ctx.heads = heads


def _attn_backward(ctx, da, daux):
    qkv, aux, a = ctx.saved_tensors
    return torch.ops.jg355.attention_core_bwd(qkv, aux, a, da.contiguous(), ctx.heads), None


_op_attention_core.register_autograd(_attn_backward, setup_context=_attn_setup)


@torch.library.custom_op("jg355::resample2", mutates_args=())
def _op_resample2(x: torch.Tensor, up: bool, scale: float) -> torch.Tensor:
    """up: nearest x2 upsample times `scale`; else 2x2 SUM pool times `scale` (avg_pool = 0.25).  Each is the other's adjoint."""
    return _up(x.contiguous(), scale) if up else _pool(x.contiguous(), scale)


@_op_resample2.register_fake
def _(x, up, scale):
    B, H, W, Cc = x.shape
    return x.new_empty((B, H * 2, W * 2, Cc)) if up else x.new_empty((B, H // 2, W // 2, Cc))


def _resample_setup(ctx, inputs, output):
    ctx.cfg = (inputs[1], inputs[2])


def _resample_backward(ctx, dy):
    up, scale = ctx.cfg
    return torch.ops.jg355.resample2(dy.contiguous(), not up, scale), None, None


_op_resample2.register_autograd(_resample_backward, setup_context=_resample_setup)


@torch.library.custom_op("jg355::linear_act", mutates_args=())
def _op_linear_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int) -> torch.Tensor:
    """y = act(x) @ weight.T + bias, fp32 (`emb_layers` / `cond_embed` of the UNet: nn.Linear behind an optional SiLU)"""
    x, weight = x.contiguous(), weight.contiguous()
    Bn, K = x.shape
    N = weight.shape[0]
    y = torch.empty((Bn, N), device=x.device, dtype=torch.float32)
    check(_lib.lib().jg_linear_fwd(x.data_ptr(), weight.data_ptr(), _p(bias), y.data_ptr(), Bn, K, N, act, _st()), "jg_linear_fwd")
    return y


@_op_linear_act.register_fake
def _(x, weight, bias, act):
    return x.new_empty((x.shape[0], weight.shape[0]), dtype=torch.float32)


@torch.library.custom_op("jg355::linear_act_bwd", mutates_args=())
def _op_linear_act_bwd(x: torch.Tensor, weight: torch.Tensor, dy: torch.Tensor, act: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    x, weight, dy = x.contiguous(), weight.contiguous(), dy.contiguous()
    Bn, K = x.shape
    N = weight.shape[0]
    dx, dW, db = torch.empty_like(x), torch.zeros_like(weight), torch.zeros((N,), device=x.device, dtype=torch.float32)
    check(_lib.lib().jg_linear_bwd(x.data_ptr(), weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), dW.data_ptr(), db.data_ptr(), Bn, K, N, act,
                                   _st()), "jg_linear_bwd")
    return dx, dW, db


@_op_linear_act_bwd.register_fake
def _(x, weight, dy, act):
    return torch.empty_like(x), torch.empty_like(weight), x.new_empty((weight.shape[0],), dtype=torch.float32)


def _linear_setup(ctx, inputs, output):
    x, weight, bias, act = inputs
    ctx.save_for_backward(x, weight)
    ctx.act, ctx.has_bias = act, bias is not None


def _linear_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dW, db = torch.ops.jg355.linear_act_bwd(x, weight, dy.contiguous(), ctx.act)
    return (dx if ctx.needs_input_grad[0] else None, dW if ctx.needs_input_grad[1] else None,
            db if (ctx.has_bias and ctx.needs_input_grad[2]) else None, None)


_op_linear_act.register_autograd(_linear_backward, setup_context=_linear_setup)


@torch.library.custom_op("jg355::gamma_embedding", mutates_args=())
def _op_gamma_embedding(gammas: torch.Tensor, dim: int, max_period: float) -> torch.Tensor:
    """sinusoidal noise-level embedding (models/modules/diffusion_utils.py:8-42); not differentiated (the levels are data)"""
    global _IN_OP
    prev, _IN_OP = _IN_OP, True
    try:
        return gamma_embedding(gammas, dim, max_period)
    finally:
        _IN_OP = prev


@_op_gamma_embedding.register_fake
def _(gammas, dim, max_period):
    return gammas.new_empty((gammas.numel(), dim), dtype=torch.float32)


@torch.library.custom_op("jg355::ddpm_prepare", mutates_args=())
def _op_ddpm_prepare(y0: torch.Tensor, ycond: torch.Tensor, noise: torch.Tensor, mask: Optional[torch.Tensor], gammas: torch.Tensor,
                     fp16: bool, cpad: int) -> torch.Tensor:
    """q_sample + mask blend + cat([y_cond, y_noisy]) of DiffusionGenerator.forward (diffusion_generator.py:480-491) -> NHWC 16-bit"""
    global _IN_OP
    prev, _IN_OP = _IN_OP, True
    try:
        return ddpm_prepare(y0, ycond, noise, mask, gammas, torch.float16 if fp16 else torch.bfloat16, cpad)
    finally:
        _IN_OP = prev


@_op_ddpm_prepare.register_fake
def _(y0, ycond, noise, mask, gammas, fp16, cpad):
    B, _, H, W = y0.shape
    return y0.new_empty((B, H, W, cpad), dtype=torch.float16 if fp16 else torch.bfloat16)


@torch.library.custom_op("jg355::ddpm_mse_loss", mutates_args=())
def _op_ddpm_mse_loss(nh: torch.Tensor, noise: torch.Tensor, mask: Optional[torch.Tensor], w: Optional[torch.Tensor], lam: float,
                      grad_scale: float, channels: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(lambda MSE(w m noise, w m noise_hat), its gradient w.r.t. noise_hat times grad_scale) in one pass (palette_model.py:597-620)"""
    B, H, W, cpad = nh.shape
    nh = nh.contiguous()
    loss = torch.zeros((), device=nh.device, dtype=torch.float32)
    dnh = torch.empty_like(nh)
    check(_lib.lib().jg_ddpm_mse_loss(_dt(nh), noise.contiguous().data_ptr(), nh.data_ptr(), _p(mask), _p(w), loss.data_ptr(), dnh.data_ptr(), B,
                                      channels, H, W, cpad, lam, grad_scale, _st()), "jg_ddpm_mse_loss")
    return loss, dnh


@_op_ddpm_mse_loss.register_fake
def _(nh, noise, mask, w, lam, grad_scale, channels):
    return nh.new_empty((), dtype=torch.float32), torch.empty_like(nh)


def _mse_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.set_materialize_grads(False)


def _mse_backward(ctx, gloss, gdnh):
    (dnh,) = ctx.saved_tensors
    g = None if gloss is None else (dnh.float() * gloss.float()).to(dnh.dtype)
    return g, None, None, None, None, None, None


_op_ddpm_mse_loss.register_autograd(_mse_backward, setup_context=_mse_setup)



from . import ops_library_cut  # noqa: E402,F401  (registers the CUT-family torch.ops.jg355.* entries)

"""Flat parameter arena: every parameter of a network lives in ONE fp32 device buffer, with
matching flat buffers for gradients, Adam moments and the EMA copy, plus 16-bit working copies
of the convolution weights in the layouts the MFMA kernels read.

Why (MI355X-first, SURVEY.md 5 "Distributed comm backend" / 8(e)):
  * one fused AdamW+EMA+zero_grad launch per step instead of 324 x (several) tensor ops;
  * one RCCL all-reduce over the flat gradient per step (chunked so the optimizer of chunk i
    overlaps the all-reduce of chunk i+1) instead of DDP's bucketed reducer;
  * wgrad / norm / linear kernels accumulate straight into the gradient arena.

nn.Parameters stay the public view (`state_dict()` keys and logical shapes are the reference's):
each `param.data` / `param.grad` is a view into the arena.  Conv2d weights are logical OIHW with
channels_last strides, i.e. physical [Cout][R][S][Cin] -- what the kernels want.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import check
from .ops import ConvMeta, _st

ALIGN = 64  # floats


def _pad8(n):
    return (n + 7) // 8 * 8


class ParamArena:
    def __init__(self, module: nn.Module, device, act_dtype, priority=("emb_layers.1.weight", "emb_layers.1.bias"), frozen_prefixes=()):
        """`frozen_prefixes`: parameter-name prefixes of sub-networks that never train (the projected discriminator's feature network).  Their
        16-bit working copies are re-derived only after a load (not after every optimizer step), and when they form the LEADING block of the
        arena the fused optimizer starts behind them -- torch.optim skips parameters without a gradient (no moment update, no weight decay), which
        is what the reference's optimizer does with them."""
        from .modules.layers import JGConvNd  # local import (layers imports ops)

        self.module = module
        self.device = torch.device(device)
        self.act_dtype = act_dtype
        named = list(module.named_parameters())
        # priority groups first, each group contiguous and in module order (lets UNet run all
        # ResBlock embedding projections as ONE linear over a stacked [sum(2C), emb] view)
        ordered, used = [], set()
        self.groups = OrderedDict()
        for suffix in priority:
            grp = [(n, p) for n, p in named if n.endswith(suffix)]
            self.groups[suffix] = [n for n, _ in grp]
            ordered += grp
            used |= {n for n, _ in grp}
        ordered += [(n, p) for n, p in named if n not in used]

        tight = set(n for names in self.groups.values() for n in names)
        cursor, self.slices = 0, OrderedDict()
        for name, p in ordered:
            n = p.numel()
            self.slices[name] = (cursor, n)
            cursor += n if name in tight else (n + ALIGN - 1) // ALIGN * ALIGN
            if name in tight and name == self._last_of_group(name):
                cursor = (cursor + ALIGN - 1) // ALIGN * ALIGN
        self.numel = max(cursor, ALIGN)
        f32 = dict(device=self.device, dtype=torch.float32)
        self.p = torch.zeros(self.numel, **f32)
        self.g = torch.zeros(self.numel, **f32)
        self.m = torch.zeros(self.numel, **f32)
        self.v = torch.zeros(self.numel, **f32)
        self.ema = None
        self.step = 0
        self.optim_kind = None     # jg_optim_step kind (0 adam, 1 adamw, 2 radam, 3 lion); None: adam / adamw by `decoupled`
        self.overflow = None       # device int32 [found non-finite in this step's gradient, steps dropped so far] (fp16 only)

        for name, prm in ordered:
            off, n = self.slices[name]
            view, gview = self._views(self.p, off, prm.shape), self._views(self.g, off, prm.shape)
            with torch.no_grad():
                view.copy_(prm.detach().to(self.device, torch.float32))
            prm.data = view
            prm.grad = gview
        for mod in module.modules():
            for k, b in list(mod._buffers.items()):
                if b is not None:
                    mod._buffers[k] = b.to(self.device)

        # ---- 16-bit working copies of the conv weights --------------------------------------
        convs = [(n, m) for n, m in module.named_modules() if isinstance(m, JGConvNd)]
        n16 = n16t = 0
        desc = []
        for name, conv in convs:
            wt = getattr(conv, conv.jg_wname)
            cout, cin = wt.shape[0], wt.shape[1]
            rs = wt.numel() // (cout * cin)
            R = S = int(round(rs ** 0.5))
            assert R * S == rs
            coutp, cinp = _pad8(cout), _pad8(cin)
            size = coutp * rs * cinp
            off, _ = self.slices[(name + "." if name else "") + conv.jg_wname]
            dst, dstT = n16, (n16t if conv.needs_dgrad else -1)
            desc.append([off, dst, dstT, cout, rs, cin, coutp, cinp])
            n16 += size
            if conv.needs_dgrad:
                n16t += size
        self.w16 = torch.zeros(max(n16, 8), device=self.device, dtype=act_dtype)
        self.w16T = torch.zeros(max(n16t, 8), device=self.device, dtype=act_dtype)
        is_frozen = [bool(frozen_prefixes) and name.startswith(tuple(frozen_prefixes)) for name, _ in convs]
        live = [d for d, f in zip(desc, is_frozen) if not f]
        froz = [d for d, f in zip(desc, is_frozen) if f]
        self.desc = torch.tensor(live, dtype=torch.int64, device=self.device).contiguous() if live else None
        self.desc_frozen = torch.tensor(froz, dtype=torch.int64, device=self.device).contiguous() if froz else None
        self._frozen_dirty = True
        # leading block of never-trained parameters: the optimizer starts behind it
        self.frozen_end = 0
        if frozen_prefixes:
            for name, prm in ordered:
                if not name.startswith(tuple(frozen_prefixes)):
                    break
                off, n = self.slices[name]
                self.frozen_end = (off + n + ALIGN - 1) // ALIGN * ALIGN
        # frozen parameters OUTSIDE the leading block would still be stepped by the fused optimizer (weight decay moves them) while their 16-bit
        # working copies are no longer refreshed: refuse the layout instead of drifting silently
        if frozen_prefixes:
            stray = [name for name, _ in ordered if name.startswith(tuple(frozen_prefixes)) and self.slices[name][0] >= self.frozen_end]
            if stray:
                raise ValueError(f"ParamArena: frozen parameters must form the leading block of the arena; behind trained ones: {stray[:4]}")
        self._bias_pads = []
        for (name, conv), d in zip(convs, desc):
            _, dst, dstT, cout, rs, cin, coutp, cinp = d
            R = S = int(round(rs ** 0.5))
            m = ConvMeta()
            m.Cin, m.Cout, m.Cin_real, m.Cout_real = cinp, coutp, cin, cout
            m.R, m.S, m.pad, m.stride = R, S, conv.jg_padding, conv.jg_stride
            m.w16 = self.w16[dst:dst + coutp * rs * cinp].view(coutp, R, S, cinp)
            m.w16T = self.w16T[dstT:dstT + coutp * rs * cinp].view(cinp, R, S, coutp) if dstT >= 0 else None
            m.weight, m.bias = getattr(conv, conv.jg_wname), getattr(conv, conv.jg_bname)
            m.bias_pad = None
            if m.bias is not None and coutp != cout:
                m.bias_pad = torch.zeros(coutp, **f32)
                self._bias_pads.append((m.bias_pad, m.bias, cout))
            conv.meta = m
        self._dirty = True
        module._jg_arena = self
        module.register_state_dict_post_hook(_state_dict_contiguous_hook)
        module.register_load_state_dict_post_hook(_mark_dirty_hook)

    def _last_of_group(self, name):
        for names in self.groups.values():
            if name in names:
                return names[-1]
        return None

    @staticmethod
    def _views(flat, off, shape):
        n = 1
        for s in shape:
            n *= s
        t = flat[off:off + n]
        if len(shape) == 4:
            O, I, R, S = shape
            return t.view(O, R, S, I).permute(0, 3, 1, 2)
        return t.view(shape)

    def group_view(self, suffix, flat=None):
        """Stacked 2-D/1-D view over a priority group (all tensors share trailing dims)."""
        names = self.groups[suffix]
        flat = self.p if flat is None else flat
        off0, _ = self.slices[names[0]]
        offl, nl = self.slices[names[-1]]
        return flat[off0:offl + nl]

    # ---- weights -------------------------------------------------------------------------
    # `arena.dirty = True` from outside (a load, a test that writes parameters) invalidates EVERY working copy; the optimizer step invalidates the
    # trained ones only (`_dirty`)
    @property
    def dirty(self):
        return self._dirty

    @dirty.setter
    def dirty(self, v):
        self._dirty = bool(v)
        if v:
            self._frozen_dirty = True

    def refresh(self):
        """Re-derive the 16-bit conv weights from the fp32 masters (after an optimizer step or a load)."""
        dt = _lib.JG_F16 if self.act_dtype == torch.float16 else _lib.JG_BF16
        if self.desc is not None:
            check(_lib.lib().jg_refresh_weights(dt, self.p.data_ptr(), self.w16.data_ptr(), self.w16T.data_ptr(),
                                                self.desc.data_ptr(), self.desc.shape[0], _st()), "jg_refresh_weights")
        if self._frozen_dirty and self.desc_frozen is not None:
            check(_lib.lib().jg_refresh_weights(dt, self.p.data_ptr(), self.w16.data_ptr(), self.w16T.data_ptr(),
                                                self.desc_frozen.data_ptr(), self.desc_frozen.shape[0], _st()), "jg_refresh_weights")
        for pad, bias, n in self._bias_pads:
            pad[:n].copy_(bias.detach())
        self._dirty = False
        self._frozen_dirty = False

    def ensure_fresh(self):
        if self.dirty:
            self.refresh()

    # ---- optimizer -----------------------------------------------------------------------
    def zero_grad(self):
        self.g.zero_()

    def enable_overflow_check(self):
        """fp16 (static loss scale): a non-finite gradient drops the optimizer step like GradScaler.step (base_model.py:1268-1274)"""
        if self.overflow is None:
            self.overflow = torch.zeros(2, device=self.device, dtype=torch.int32)

    def check_overflow(self):
        """scan the (already reduced) gradient arena for inf / NaN into overflow[0]; the optimizer launches of this step read it"""
        if self.overflow is None:
            return
        self.overflow[:1].zero_()
        check(_lib.lib().jg_grad_nonfinite(self.g.data_ptr(), self.numel, self.overflow.data_ptr(), _st()), "jg_grad_nonfinite")

    def adamw_step(self, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale=1.0, ema_beta=None, zero_grad=True,
                   lo=0, hi=None):
        """One fused optimizer(+EMA)(+zero_grad) launch over arena[lo:hi] (Adam / AdamW / RAdam / Lion by `optim_kind`);
        `self.step` must already be advanced."""
        hi = self.numel if hi is None else hi
        lo = max(lo, self.frozen_end)          # never-trained leading block: untouched, like parameters without a gradient in torch.optim
        if lo >= hi:
            self._dirty = True
            return
        ema_ptr = None
        if ema_beta is not None:
            if self.ema is None:
                raise RuntimeError("EMA buffer not created")
            ema_ptr = self.ema.data_ptr() + 4 * lo
        kind = self.optim_kind if self.optim_kind is not None else int(bool(decoupled))
        skip = nsk = None
        if self.overflow is not None:
            skip, nsk = self.overflow.data_ptr(), (self.overflow.data_ptr() + 4 if lo == 0 else None)   # count a dropped step once
        check(_lib.lib().jg_optim_step(kind, self.p.data_ptr() + 4 * lo, self.g.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo,
                                       self.v.data_ptr() + 4 * lo, ema_ptr, hi - lo, lr, beta1, beta2, eps, weight_decay,
                                       self.step, grad_scale, 0.0 if ema_beta is None else ema_beta, int(zero_grad), skip, nsk,
                                       _st()), "jg_optim_step")
        self._dirty = True

    def ema_create(self):
        self.ema = self.p.clone()

    def ema_update(self, beta):
        check(_lib.lib().jg_ema_update(self.ema.data_ptr(), self.p.data_ptr(), self.numel, beta, _st()), "jg_ema_update")

    def named_views(self, flat):
        """name -> view of `flat` (same layout as the parameters), e.g. the EMA copy."""
        out = OrderedDict()
        for name, prm in self.module.named_parameters():
            off, _ = self.slices[name]
            out[name] = self._views(flat, off, prm.shape)
        return out


def _state_dict_contiguous_hook(module, state_dict, prefix, local_metadata):
    # checkpoints hold plain contiguous tensors in the reference's logical layout
    for k, v in list(state_dict.items()):
        if k.startswith(prefix) and torch.is_tensor(v):
            state_dict[k] = v.detach().clone(memory_format=torch.contiguous_format)


def _mark_dirty_hook(module, incompatible_keys):
    arena = getattr(module, "_jg_arena", None)
    if arena is not None:
        arena.dirty = True

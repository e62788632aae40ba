"""Fused AdamW/Adam over a ParamArena, exposed as a torch.optim.Optimizer so that the
reference's scheduler code (`get_scheduler`, models/modules/utils.py:115-157) and
`optimizer.param_groups[0]["lr"]` keep working.  Replaces `train.optim` (train.py:51-62) for
`train_optim in {"adam", "adamw"}`."""
from __future__ import annotations

import torch

from . import parallel


KINDS = {"adam": 0, "adamw": 1, "radam": 2, "lion": 3}


class FusedAdamW(torch.optim.Optimizer):
    """`kind`: "adam" | "adamw" | "radam" | "lion" (train.py:51-62; adam8bit needs bitsandbytes' quantisation maps: not provided)."""

    def __init__(self, arena, params, lr, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8, decoupled=True, kind=None):
        defaults = dict(lr=lr, betas=betas, weight_decay=weight_decay, eps=eps)
        super().__init__(list(params), defaults)
        self.arena = arena
        self.decoupled = decoupled
        self.kind = kind if kind is not None else ("adamw" if decoupled else "adam")
        arena.optim_kind = KINDS[self.kind]
        self.grad_scale = 1.0      # 1 / (loss_scale) ; the data-parallel 1/world is applied by parallel.py
        self.n_chunks = 4          # all-reduce / optimizer overlap granularity when world_size > 1

    @torch.no_grad()
    def step(self, closure=None, ema_beta=None):
        """One launch over the whole arena (or n_chunks launches pipelined behind the chunked
        gradient all-reduce).  Also zeroes the gradients (fused zero_grad) and, when `ema_beta`
        is given, updates the EMA copy in the same pass."""
        g = self.param_groups[0]
        a = self.arena
        a.step += 1
        hp = dict(lr=float(g["lr"]), beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"],
                  weight_decay=g["weight_decay"], decoupled=self.decoupled, ema_beta=ema_beta, zero_grad=True)
        if parallel.exchange_active():
            parallel.allreduce_and_step(a, hp, self.grad_scale, self.n_chunks)
        else:
            a.check_overflow()
            a.adamw_step(grad_scale=self.grad_scale, **hp)
        a.refresh()

    def zero_grad(self, set_to_none=False):
        # gradients are permanent arena views; step() already zeroed them
        self.arena.zero_grad()

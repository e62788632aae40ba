"""Pure data parallelism over the 8 GPUs of one node: one process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI), full replica per rank, and ONE exchange
per optimizer step: an all-reduce(SUM) of the flat gradient arena, issued in a few large chunks
so that the fused AdamW of chunk i runs while chunk i+1 is still on the wire.  Replaces the
reference's per-network DistributedDataParallel wrappers (models/base_model.py:725-737) and its
`no_sync()` accumulation contexts (:1313-1315)."""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist

_NO_SYNC = 0
# run the chunked all-reduce + optimizer pipeline even with a single rank (tests exercise the RCCL path on a 1-GPU box)
FORCE_EXCHANGE = False


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def in_no_sync():
    return _NO_SYNC > 0


def exchange_active():
    """True when optimizer steps must all-reduce the flat gradient first."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_EXCHANGE) and not in_no_sync()


@contextlib.contextmanager
def no_sync():
    """Gradient-accumulation micro-steps: gradients stay local (arena accumulates)."""
    global _NO_SYNC
    _NO_SYNC += 1
    try:
        yield
    finally:
        _NO_SYNC -= 1


def chunk_bounds(n, n_chunks, align=1024):
    per = (n + n_chunks - 1) // n_chunks
    per = (per + align - 1) // align * align
    return [(lo, min(n, lo + per)) for lo in range(0, n, per)]


def allreduce_and_step(arena, hp, grad_scale, n_chunks=4):
    """sum-all-reduce arena.g chunk by chunk (async, RCCL stream) and run the fused optimizer on
    each chunk as soon as its reduction has landed.  Mean over ranks = DDP semantics."""
    ws = world_size()
    bounds = chunk_bounds(arena.numel, n_chunks)
    works = [dist.all_reduce(arena.g[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for lo, hi in bounds]
    for (lo, hi), w in zip(bounds, works):
        w.wait()  # NCCL/RCCL: makes the current stream wait, does not block the host
        arena.adamw_step(grad_scale=grad_scale / ws, lo=lo, hi=hi, **hp)


def broadcast_params(arena, src=0):
    """DDP's constructor broadcast: every rank starts from rank `src`'s parameters."""
    if world_size() > 1:
        dist.broadcast(arena.p, src)
        arena.dirty = True


class FlatDataParallel(torch.nn.Module):
    """`net.module` / `net.no_sync()` surface of DistributedDataParallel for code written against
    the reference (models/base_model.py:836-860,1313-1315); forward just delegates -- the
    gradient exchange happens in FusedAdamW.step()."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def no_sync(self):
        return no_sync()

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)

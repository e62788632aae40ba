"""Pure data parallelism over the 8 GPUs of one node: one process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI), full replica per rank, and ONE exchange
per optimizer step: an all-reduce(SUM) of the flat gradient arena, issued in a few large chunks
so that the fused AdamW of chunk i runs while chunk i+1 is still on the wire.  Networks whose step
runs exactly ONE backward (palette / cm) additionally start the reduction of a chunk from INSIDE the
backward, as soon as every parameter of the chunk has its final gradient (`EarlyExchange`): the wire
time hides behind the remaining input-gradient / weight-gradient kernels.  Replaces the
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


# id(parameter) -> EarlyExchange that owns it (filled by EarlyExchange.__init__)
_EARLY = {}
# callables run right before a gradient chunk's all-reduce is enqueued: schedules that launch gradient kernels on a second stream
# (modules/unet_exec.py: weight gradients) register their join here, because the collective is ordered behind the current stream only
PRE_LAUNCH_HOOKS = []


def _live(hooks):
    """the callables of a hook list; entries may be weakref.WeakMethod (an executor's bound methods: a network that is gone takes its
    hooks -- and its side stream -- with it instead of living in a process-global list for ever), dead ones are dropped"""
    import weakref

    out = []
    for h in list(hooks):
        if isinstance(h, weakref.WeakMethod):
            f = h()
            if f is None:
                hooks.remove(h)
                continue
            out.append(f)
        else:
            out.append(h)
    return out
# context-manager factories entered around the LAUNCH of an early chunk (EarlyExchange._launch).  The UNet executor registers one that
# enqueues the collective from its weight-gradient stream (after making that stream wait for the compute stream): the chunk is then
# ordered behind the weight gradients WITHOUT the compute stream having to wait for them -- joining the two streams before every chunk
# (round 2) serialised exactly the overlap the second stream exists for.  Empty: the PRE_LAUNCH_HOOKS join is used instead.
LAUNCH_CONTEXTS = []
# "fp32" (default) | "bf16": wire format of the gradient chunks (JG_GRAD_WIRE).  bf16 halves the bytes on xGMI (237 -> 119 MB per step
# for the 59 M-parameter UNet): the chunk is rounded into a bf16 staging buffer, summed by RCCL in bf16 and widened back into the fp32
# gradient arena before the optimizer reads it (the optimizer state stays fp32).  The rounding is that of one more bf16 hop on a gradient
# that was computed from bf16 activations; off by default, the reference's DDP reduces fp32.
import os as _os
GRAD_WIRE = _os.environ.get("JG_GRAD_WIRE", "fp32")
# per-step diagnostics of the exchange (bench.py --gpus N prints them per rank): CUDA events around the wait + optimizer loop of
# allreduce_and_step, i.e. the time the compute stream spent on the exposed part of the exchange plus the chunked optimizer
TIMING = None       # set to [] to collect (ev_begin, ev_end) pairs


def _launch_allreduce(buf):
    """async all-reduce(SUM) of a slice of the fp32 gradient arena, in the configured wire format; returns an object with .wait()"""
    if GRAD_WIRE == "bf16" and buf.dtype == torch.float32:
        wire = buf.to(torch.bfloat16)
        work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True)
        return _WireWork(work, wire, buf)
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)


class _WireWork:
    """work handle of a reduced-precision chunk: wait() = wait for the collective, then widen the sum back into the fp32 arena slice"""

    def __init__(self, work, wire, dst):
        self.work, self.wire, self.dst = work, wire, dst

    def wait(self):
        self.work.wait()
        if self.wire.is_cuda:      # the staging buffer may have been allocated on the launching (weight-gradient) stream
            self.wire.record_stream(torch.cuda.current_stream())
        self.dst.copy_(self.wire)


class EarlyExchange:
    """Backward-overlapped reduction of one gradient arena (DDP's bucket-ready hook, on the flat arena).

    The arena is cut into `n_chunks` aligned chunks.  The backward schedule reports parameters whose gradient is
    final (`grads_final`); the moment the last parameter overlapping a chunk is reported, the chunk's
    all-reduce(SUM) is enqueued (async: the RCCL stream waits for the kernels already launched on the compute
    stream, the backward keeps launching).  `drain()` -- called by the optimizer step -- enqueues what is left
    (parameters outside the fused backward, padding-only chunks) and returns every chunk with its work handle in
    launch order.  Every rank runs the same schedule, so the collectives are issued in the same order everywhere.

    Contract: between two optimizer steps each parameter is reported at most once and its gradient is not touched
    afterwards, i.e. ONE backward per step over this arena (accumulation micro-steps run under `no_sync()` and
    report nothing).  Models opt in (`BaseModel.overlap_exchange`)."""

    def __init__(self, arena, named_params, n_chunks=8):
        self.arena = arena
        self.bounds = chunk_bounds(arena.numel, n_chunks)
        per = self.bounds[0][1] - self.bounds[0][0]
        self.owner = {}
        self.total = [0] * len(self.bounds)
        self.last_early = 0
        for name, prm in named_params:
            off, n = arena.slices[name]
            cs = tuple(range(off // per, min((off + max(n, 1) - 1) // per, len(self.bounds) - 1) + 1))
            self.owner[id(prm)] = cs
            _EARLY[id(prm)] = self
            for c in cs:
                self.total[c] += 1
        self._reset()

    def _reset(self):
        self.left = list(self.total)
        self.seen = set()
        self.launched = []          # [(chunk index, work)]
        self.sent = [False] * len(self.bounds)

    def _launch(self, c):
        lo, hi = self.bounds[c]
        self.sent[c] = True
        if _live(LAUNCH_CONTEXTS):
            with contextlib.ExitStack() as stack:
                for make in _live(LAUNCH_CONTEXTS):
                    stack.enter_context(make())
                work = _launch_allreduce(self.arena.g[lo:hi])
        else:
            for hook in _live(PRE_LAUNCH_HOOKS):
                hook()
            work = _launch_allreduce(self.arena.g[lo:hi])
        self.launched.append((c, work))

    def mark(self, prm):
        k = id(prm)
        if k in self.seen:
            raise RuntimeError("EarlyExchange: a parameter was reported final twice in one step (more than one backward over this "
                               "arena per optimizer step: the model must not opt into overlap_exchange)")
        self.seen.add(k)
        for c in self.owner[k]:
            self.left[c] -= 1
            if self.left[c] == 0:
                self._launch(c)

    def drain(self):
        self.last_early = len(self.launched)     # chunks that went out from inside the backward (diagnostics / tests)
        for c in range(len(self.bounds)):
            if not self.sent[c]:
                self._launch(c)
        out = [(self.bounds[c][0], self.bounds[c][1], w) for c, w in self.launched]
        self._reset()
        return out


def grads_final(params):
    """Called by a fused backward schedule: the gradients of `params` are final for this step."""
    if not _EARLY or not exchange_active():
        return
    for prm in params:
        ex = _EARLY.get(id(prm))
        if ex is not None:
            ex.mark(prm)


def allreduce_and_step(arena, hp, grad_scale, n_chunks=4):
    """sum-all-reduce arena.g chunk by chunk (async, RCCL stream) and run the fused optimizer on
    each chunk as soon as its reduction has landed.  Mean over ranks = DDP semantics.
    Chunks whose reduction was already started from inside the backward (`EarlyExchange`) are only waited for."""
    ws = world_size()
    overflow = getattr(arena, "overflow", None)
    # every gradient kernel launched on another stream is joined FIRST: from here on the whole arena is final on the current stream
    # (the overflow scan below reads all of it, and the chunks that have not left yet are enqueued behind the current stream)
    for hook in _live(PRE_LAUNCH_HOOKS):
        hook()
    ex = getattr(arena, "early_exchange", None)
    if overflow is not None and (ex is None or not ex.launched):
        # fp16: the step is dropped on every rank or on none.  No chunk has left yet: the LOCAL gradient is scanned before it is
        # reduced and the flag travels as one tiny MAX all-reduce.
        arena.check_overflow()
        dist.all_reduce(overflow[:1], op=dist.ReduceOp.MAX)
        overflow = None
    if ex is not None:
        pending = ex.drain()
    else:
        pending = [(lo, hi, _launch_allreduce(arena.g[lo:hi])) for lo, hi in chunk_bounds(arena.numel, n_chunks)]
    timed = TIMING is not None and torch.cuda.is_available() and arena.g.is_cuda
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if overflow is not None:
        # chunks went out from inside the backward (EarlyExchange): some slices are being summed in place right now, so the scan runs
        # on the REDUCED arena, after every chunk has landed -- a non-finite value of any rank survives the sum, every rank scans the
        # same reduced values and takes the same decision without a flag exchange.  The optimizer chunks then run back to back.
        for lo, hi, w in pending:
            w.wait()
        arena.check_overflow()
        for lo, hi, w in pending:
            arena.adamw_step(grad_scale=grad_scale / ws, lo=lo, hi=hi, **hp)
    else:
        for lo, hi, w in pending:
            w.wait()  # NCCL/RCCL: makes the current stream wait, does not block the host
            arena.adamw_step(grad_scale=grad_scale / ws, lo=lo, hi=hi, **hp)
    if timed:
        ev1.record()
        TIMING.append((ev0, ev1))


def broadcast_params(arena, src=0):
    """DDP's constructor broadcast: every rank starts from rank `src`'s parameters AND buffers (BatchNorm running statistics,
    spectral-norm vectors, noise schedules: `_sync_module_states` covers both in DistributedDataParallel)."""
    if world_size() > 1:
        dist.broadcast(arena.p, src)
        arena.dirty = True
        module = getattr(arena, "module", None)
        if module is not None:
            for b in module.buffers():
                if b is not None and b.numel() > 0:
                    dist.broadcast(b, src)


def reduce_losses(losses):
    """train.py:293-301 of the reference: before printing, every logged loss is all-reduced (SUM) over the ranks and divided by the
    world size.  The reference issues one collective per loss; here the values are stacked and travel as ONE small all-reduce.
    `losses`: OrderedDict name -> 0-dim tensor | float (BaseModel.get_current_losses()); returns the same mapping with the mean over
    ranks (0-dim fp32 tensors on the first tensor's device).  Single process: returned unchanged."""
    ws = world_size()
    if ws == 1 or not losses:
        return losses
    dev = next((v.device for v in losses.values() if torch.is_tensor(v)), torch.device("cpu"))
    flat = torch.stack([(v.detach().float().to(dev) if torch.is_tensor(v) else torch.tensor(float(v), device=dev)).reshape(()) for v in losses.values()])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= ws
    return type(losses)((k, flat[i]) for i, k in enumerate(losses))


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

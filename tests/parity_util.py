"""Shared helpers of the GPU parity tests: teacher forcing and single-step update checks.

Trajectory comparisons after the first optimizer step are ill-posed under 16-bit activations (Adam's first steps are sign-like,
so the rounding noise of a near-zero gradient moves that weight by a full `lr`): two correct implementations separate, and a
bound fitted to one box's separation is flaky on the next (atomics make the summation order run-dependent).  The step tests
therefore TEACHER-FORCE: before every iteration the HIP model receives the CPU oracle's complete state (weights, buffers, Adam
moments, step count, EMA), both sides run ONE `optimize_parameters()` on the same injected randomness, and what is compared is
  * the losses of that iteration (forward quantities of identical weights -> a pure forward tolerance), and
  * the parameter UPDATE of that iteration against the oracle's update (cosine + norm ratio): a missing, zero or mis-signed
    update fails, which the former `|norm| +- travel` checks could not see.
"""
import torch


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def force_state(net, params, m=None, v=None, step=None, ema=None, buffers=None):
    """net (arena-backed module) := the oracle's state.  `params` / `m` / `v` / `ema`: name -> fp32 CPU tensor in the reference's
    logical layout; `buffers`: extra state_dict entries (BatchNorm running statistics)."""
    arena = net.arena
    sd = net.state_dict()
    for k, t in params.items():
        assert k in sd, k
        sd[k] = t
    for k, t in (buffers or {}).items():
        assert k in sd, k
        sd[k] = t
    net.load_state_dict(sd)
    dev = arena.p.device
    with torch.no_grad():
        for flat, src in ((arena.m, m), (arena.v, v), (arena.ema, ema)):
            if src is None:
                continue
            assert flat is not None, "the HIP side has no EMA buffer yet"
            views = arena.named_views(flat)
            for k, t in src.items():
                views[k].copy_(t.to(dev))
    if step is not None:
        arena.step = int(step)
    arena.g.zero_()
    arena.refresh()


def snapshot(net):
    return {k: p.detach().float().cpu().clone() for k, p in net.named_parameters()}


def update_agreement(before, after, ref_before, ref_after, skip=lambda k: False):
    """cosine and norm ratio between this implementation's update (after - before) and the oracle's, over all parameters not
    excluded by `skip` (concatenated), plus the per-tensor table [(cos, ratio, numel, name)]."""
    num = da = db = 0.0
    table = []
    for k in ref_after:
        if skip(k):
            continue
        a = (after[k].double() - before[k].double()).flatten()
        b = (ref_after[k].double() - ref_before[k].double()).flatten()
        n, na, nb = float(a @ b), float(a @ a), float(b @ b)
        num, da, db = num + n, da + na, db + nb
        table.append((n / ((na * nb) ** 0.5 + 1e-300), (na / (nb + 1e-300)) ** 0.5, a.numel(), k))
    return num / ((da * db) ** 0.5 + 1e-300), (da / (db + 1e-300)) ** 0.5, table


def check_update(tag, before, after, ref_before, ref_after, cos_min, skip=lambda k: False, log=None):
    """A correct single Adam(W) step from identical (w, m, v, t): same direction as the oracle's (global cosine >= cos_min), same
    length (every element moves by <= ~lr, so the ratio is tight unless part of the update is missing), every sizeable tensor moved
    and none of them the wrong way."""
    cos, ratio, table = update_agreement(before, after, ref_before, ref_after, skip)
    if log is not None:
        log.append(f"{tag}: cos={cos:.4f} ratio={ratio:.4f}")
        for c, r, n, k in sorted(table)[:5]:
            log.append(f"    worst {c:.3f} ratio {r:.3f} numel {n} {k}")
    assert cos >= cos_min, (tag, "update direction", cos, sorted(table)[:5])
    assert 0.9 < ratio < 1.1, (tag, "update length", ratio)
    for c, r, n, k in table:
        if n >= 512:
            assert r > 0.5, (tag, "tensor did not move", k, r)
            assert c > 0.5, (tag, "tensor moved across / against the oracle's update", k, c)      # observed >= 0.77 (CUT), >= 0.9 (palette / cm)
    return cos, ratio


def check_ema(tag, ema_before, ema_after, p_after, beta, first):
    """ema_step (base_model.py:1284-1297): the first call copies the parameters, later calls are p + beta * (ema - p)."""
    for k, e in ema_after.items():
        want = p_after[k] if first else p_after[k] + beta * (ema_before[k] - p_after[k])
        assert relerr(e, want) < 1e-6, (tag, k, relerr(e, want))

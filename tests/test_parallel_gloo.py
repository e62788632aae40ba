"""CPU, world_size 2 (gloo): the data-parallel exchange of joligen_amd/parallel.py -- chunked
all-reduce of the flat gradient + per-chunk optimizer, mean-over-ranks semantics, no_sync
accumulation, parameter broadcast -- against a single-process computation on the summed batch.
The GPU kernels are replaced by a torch stand-in with the same `adamw_step(lo, hi)` contract."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import jg_oracle as O


class FakeArena:
    """Same duck-typed surface as ParamArena, CPU math from the oracle."""

    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.numel = n
        self.p = torch.randn(n, generator=g)
        self.g = torch.zeros(n)
        self.m = torch.zeros(n)
        self.v = torch.zeros(n)
        self.ema = None
        self.step = 0
        self.dirty = False
        self.calls = []

    def adamw_step(self, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale=1.0, ema_beta=None, zero_grad=True,
                   lo=0, hi=None):
        hi = self.numel if hi is None else hi
        self.calls.append((lo, hi))
        s = slice(lo, hi)
        O.adamw_step([self.p[s]], [self.g[s] * grad_scale], [self.m[s]], [self.v[s]], self.step, lr, beta1, beta2, eps,
                     weight_decay, decoupled)
        if zero_grad:
            self.g[s].zero_()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from joligen_amd import parallel

    assert parallel.world_size() == world and parallel.rank() == rank
    arena = FakeArena(n, seed=100 + rank)       # ranks start different ...
    parallel.broadcast_params(arena, 0)         # ... and agree after the broadcast
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=True, ema_beta=None, zero_grad=True)
    g = torch.Generator().manual_seed(7)
    grads = [torch.randn(world, n, generator=g) for _ in range(3)]
    # step 1: plain step
    arena.g += grads[0][rank]
    arena.step += 1
    parallel.allreduce_and_step(arena, hp, grad_scale=1.0, n_chunks=4)
    # step 2: two accumulation micro-steps; the first one is local (no_sync), then one exchange
    with parallel.no_sync():
        assert parallel.in_no_sync()
        arena.g += 0.5 * grads[1][rank]
    assert not parallel.in_no_sync()
    arena.g += 0.5 * grads[2][rank]
    arena.step += 1
    parallel.allreduce_and_step(arena, hp, grad_scale=1.0, n_chunks=3)
    torch.save(dict(p=arena.p, calls=arena.calls), out % rank)
    dist.destroy_process_group()


def test_flat_allreduce_matches_single_process_mean(tmp_path):
    world, n = 2, 5000
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert torch.equal(r0["p"], r1["p"])                       # replicas stay identical
    # reference: one process, gradient = mean over ranks
    ref = FakeArena(n, seed=100)
    g = torch.Generator().manual_seed(7)
    grads = [torch.randn(world, n, generator=g) for _ in range(3)]
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=True)
    ref.g += grads[0].mean(0)
    ref.step += 1
    ref.adamw_step(**hp)
    ref.g += 0.5 * grads[1].mean(0) + 0.5 * grads[2].mean(0)
    ref.step += 1
    ref.adamw_step(**hp)
    torch.testing.assert_close(r0["p"], ref.p, rtol=1e-5, atol=1e-6)
    # the arena was covered exactly once per step, in chunk order
    from joligen_amd.parallel import chunk_bounds

    assert r0["calls"] == chunk_bounds(n, 4) + chunk_bounds(n, 3)
    for nch in (1, 3, 4, 7):
        b = chunk_bounds(n, nch)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))


def _worker_multi(rank, world, port, sizes, out):
    """the CUT step has three arenas (G, F, D): two are stepped back to back after one backward, the third later in the iteration;
    tiny arenas (F's MLPs, a few hundred floats in small configs) are a single ragged chunk"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from joligen_amd import parallel

    arenas = [FakeArena(n, seed=10 * i + rank) for i, n in enumerate(sizes)]
    for a in arenas:
        parallel.broadcast_params(a, 0)
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False, ema_beta=None, zero_grad=True)
    for it in range(2):
        g = torch.Generator().manual_seed(50 + it)
        grads = [torch.randn(world, n, generator=g) for n in sizes]
        for i in (0, 1):                       # group G: generator and feature network
            arenas[i].g += grads[i][rank]
            arenas[i].step += 1
            parallel.allreduce_and_step(arenas[i], hp, grad_scale=1.0, n_chunks=4)
        arenas[2].g += grads[2][rank]          # group D
        arenas[2].step += 1
        parallel.allreduce_and_step(arenas[2], hp, grad_scale=1.0, n_chunks=4)
    torch.save([a.p for a in arenas], out % rank)
    dist.destroy_process_group()


def test_three_arenas_of_the_cut_step(tmp_path):
    world, sizes = 2, (70000, 300, 5000)
    out = str(tmp_path / "m%d.pt")
    mp.spawn(_worker_multi, args=(world, _free_port(), sizes, out), nprocs=world, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    refs = [FakeArena(n, seed=10 * i) for i, n in enumerate(sizes)]
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False)
    for it in range(2):
        g = torch.Generator().manual_seed(50 + it)
        grads = [torch.randn(world, n, generator=g) for n in sizes]
        for i in range(3):
            refs[i].g += grads[i].mean(0)
            refs[i].step += 1
            refs[i].adamw_step(**hp)
    for a, b, ref in zip(r0, r1, refs):
        assert torch.equal(a, b)
        torch.testing.assert_close(a, ref.p, rtol=1e-5, atol=1e-6)


class _P:
    """stand-in for an nn.Parameter (EarlyExchange keys parameters by identity)"""


def _worker_early(rank, world, port, sizes, out):
    """backward-overlapped exchange: parameters are reported final tail-first, group by group, while the 'backward' is still
    writing the gradients of earlier parameters; two parameters (the embedding projections of the real model) are never
    reported and go out with the optimizer step"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from joligen_amd import parallel

    n = sum(sizes)
    arena = FakeArena(n, seed=100 + rank)
    params, off = [], 0
    arena.slices = {}
    for i, sz in enumerate(sizes):
        arena.slices["p%d" % i] = (off, sz)
        params.append(("p%d" % i, _P()))
        off += sz
    parallel.broadcast_params(arena, 0)
    arena.early_exchange = ex = parallel.EarlyExchange(arena, params, n_chunks=6)
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=True, ema_beta=None, zero_grad=True)
    g = torch.Generator().manual_seed(11)
    early = []
    for step in range(3):
        grads = torch.randn(world, n, generator=g)
        if step == 1:       # an accumulation micro-step under no_sync: local, nothing is reported
            with parallel.no_sync():
                arena.g += 0.5 * grads[rank]
                parallel.grads_final([p for _, p in params])
            assert not ex.seen and not ex.launched
            grads = 0.5 * torch.randn(world, n, generator=g)
        for i in range(len(sizes) - 1, 1, -1):          # parameters 0 and 1 are never reported
            lo, sz = arena.slices["p%d" % i]
            arena.g[lo:lo + sz] += grads[rank, lo:lo + sz]
            parallel.grads_final([params[i][1]])
        for i in (0, 1):
            lo, sz = arena.slices["p%d" % i]
            arena.g[lo:lo + sz] += grads[rank, lo:lo + sz]
        early.append(len(ex.launched))
        arena.step += 1
        parallel.allreduce_and_step(arena, hp, grad_scale=1.0)
        assert ex.last_early == early[-1] and not ex.launched and not ex.seen
    # reporting a parameter twice in one step is a contract violation
    parallel.grads_final([params[-1][1]])
    try:
        parallel.grads_final([params[-1][1]])
        dup = False
    except RuntimeError:
        dup = True
    ex.drain()
    torch.save(dict(p=arena.p, calls=arena.calls, early=early, dup=dup, nchunks=len(ex.bounds)), out % rank)
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 4])
def test_backward_overlapped_exchange(tmp_path, world):
    sizes = (900, 1500, 3000, 40, 5000, 7000, 64, 2048, 6000, 1000)
    n = sum(sizes)
    out = str(tmp_path / "e%d.pt")
    mp.spawn(_worker_early, args=(world, _free_port(), sizes, out), nprocs=world, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert torch.equal(r0["p"], r1["p"])
    assert r0["dup"] and r1["dup"]
    assert r0["early"] == r1["early"] and min(r0["early"]) >= r0["nchunks"] - 2 and max(r0["early"]) < r0["nchunks"]
    # every step covers the arena exactly once (chunks in launch order: tail first)
    per_step = len(r0["calls"]) // 3
    for s in range(3):
        c = sorted(r0["calls"][s * per_step:(s + 1) * per_step])
        assert c[0][0] == 0 and c[-1][1] == n and all(c[i][1] == c[i + 1][0] for i in range(len(c) - 1))
    assert r0["calls"][0][1] == n       # the tail chunk went first
    ref = FakeArena(n, seed=100)
    g = torch.Generator().manual_seed(11)
    hp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=True)
    for step in range(3):
        grads = torch.randn(world, n, generator=g)
        if step == 1:
            ref.g += 0.5 * grads.mean(0)
            grads = 0.5 * torch.randn(world, n, generator=g)
        ref.g += grads.mean(0)
        ref.step += 1
        ref.adamw_step(**hp)
    torch.testing.assert_close(r0["p"], ref.p, rtol=1e-5, atol=1e-6)


def _worker_losses(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from collections import OrderedDict

    from joligen_amd import parallel

    losses = OrderedDict([("G_tot", torch.tensor(1.0 + rank)), ("G_NCE", 0.5 * (rank + 1)), ("D_tot", torch.tensor(3.0 - 2 * rank))])
    red = parallel.reduce_losses(losses)
    torch.save({k: float(v) for k, v in red.items()}, out % rank)
    dist.destroy_process_group()


def test_logging_loss_allreduce(tmp_path):
    """train.py:293-301 of the reference: the printed losses are the mean over the ranks (here one stacked all-reduce); tensors and
    plain floats are both accepted, the key order is kept, and a single process gets its mapping back unchanged."""
    from collections import OrderedDict

    from joligen_amd import parallel

    out = str(tmp_path / "l%d.pt")
    mp.spawn(_worker_losses, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert r0 == r1 and list(r0) == ["G_tot", "G_NCE", "D_tot"]
    assert abs(r0["G_tot"] - 1.5) < 1e-6 and abs(r0["G_NCE"] - 0.75) < 1e-6 and abs(r0["D_tot"] - 2.0) < 1e-6
    single = OrderedDict(a=torch.tensor(2.0))
    assert parallel.reduce_losses(single) is single


# ---- model level: the palette step at 2 x B/2 through the exchange == the single-process step at B ----------------
MODEL_CFG = dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[16], efficient=True, S=32, B=4)


def _model_inputs():
    c = MODEL_CFG
    cfg = O.UNetCfg(in_channel=6, inner_channel=c["ngf"], out_channel=3, res_blocks=c["res_blocks"], attn_res=c["attn_res"],
                    channel_mults=c["mults"], efficient=c["efficient"])
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.options import opt_from_json

    ov = dict(G_ngf=c["ngf"], G_unet_mha_channel_mults=c["mults"], G_unet_mha_res_blocks=c["res_blocks"], G_unet_mha_attn_res=c["attn_res"],
              G_unet_mha_vit_efficient=c["efficient"], data_crop_size=c["S"], train_batch_size=c["B"])
    sd = O.synth_state_dict(define_G(**vars(opt_from_json({}, ov))).state_dict(), seed=0)
    g = torch.Generator().manual_seed(21)
    B, S = c["B"], c["S"]
    steps = []
    for _ in range(2):
        Bimg = torch.rand(B, 3, S, S, generator=g) * 2 - 1
        mask = torch.zeros(B, 1, S, S, dtype=torch.int64)
        mask[:, :, 6:20, 8:26] = 1
        A = Bimg * (1 - mask) + torch.randn(B, 3, S, S, generator=g) * mask
        t, u, noise = O.draw_step_randomness(g, Bimg, 2000)
        steps.append((Bimg, A, mask, noise, t, u))
    return cfg, sd, steps


class _OracleArena(FakeArena):
    """flat view of an OraclePaletteTrainer's parameters: the arena surface parallel.allreduce_and_step drives"""

    def __init__(self, tr):
        self.names = tr.param_names
        self.shapes = [tr.P[k].shape for k in self.names]
        self.numel = sum(int(torch.tensor(s).prod()) if len(s) else 1 for s in self.shapes)
        self.p = torch.cat([tr.P[k].reshape(-1) for k in self.names])
        self.g, self.m, self.v = torch.zeros(self.numel), torch.zeros(self.numel), torch.zeros(self.numel)
        self.ema, self.step, self.dirty, self.calls = None, 0, False, []

    def load_grads(self, grads):
        self.g += torch.cat([grads[k].reshape(-1) for k in self.names])

    def store_params(self, tr):
        off = 0
        for k, s in zip(self.names, self.shapes):
            n = tr.P[k].numel()
            tr.P[k] = self.p[off:off + n].reshape(s).clone()
            off += n


def _worker_model(rank, world, port, out, wire):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from joligen_amd import parallel

    parallel.GRAD_WIRE = wire
    cfg, sd, steps = _model_inputs()
    if rank == 1:      # ranks start from different weights; the constructor broadcast makes them rank 0's
        sd = {k: (v + 0.01 if torch.is_floating_point(v) and not O._is_buffer(k) else v) for k, v in sd.items()}
    tr = O.OraclePaletteTrainer(sd, cfg, ema_beta=None)
    arena = _OracleArena(tr)
    parallel.broadcast_params(arena, 0)
    arena.store_params(tr)
    hp = dict(lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=True, ema_beta=None, zero_grad=True)
    half = MODEL_CFG["B"] // world
    sl = slice(rank * half, (rank + 1) * half)
    losses = []
    for Bimg, A, mask, noise, t, u in steps:           # each rank: its half of the batch (DistributedSampler semantics)
        loss, grads, _ = tr.loss_and_grads(Bimg[sl], A[sl], mask[sl], noise[sl], t[sl], u[sl])
        arena.load_grads(grads)
        arena.step += 1
        parallel.allreduce_and_step(arena, hp, grad_scale=1.0, n_chunks=4)
        arena.store_params(tr)
        losses.append(float(parallel.reduce_losses({"G_tot": loss})["G_tot"]))
    torch.save(dict(p=arena.p, losses=losses), out % rank)
    dist.destroy_process_group()


def _single_process_reference():
    cfg, sd, steps = _model_inputs()
    tr = O.OraclePaletteTrainer(sd, cfg, ema_beta=None)
    losses = []
    for Bimg, A, mask, noise, t, u in steps:
        losses.append(float(tr.optimize_parameters(Bimg, A, mask, noise, t, u)))
    return torch.cat([tr.P[k].reshape(-1) for k in tr.param_names]), losses


def test_palette_step_two_ranks_equals_single_process_batch(tmp_path):
    """the FULL palette optimisation step (DDPM loss, UNet backward, AdamW) of a batch of 4 split over two ranks -- every rank computes
    its half with the CPU oracle (the HIP model needs a GPU), the gradients travel through parallel.allreduce_and_step in 4 chunks with
    DDP mean semantics -- reproduces the single-process step on the whole batch: the MSE loss is a mean over B*C*H*W, so the mean of
    the two half-batch gradients IS the full-batch gradient (models/base_model.py:725-737 of the reference: DDP averages)."""
    world = 2
    out = str(tmp_path / "pm%d.pt")
    mp.spawn(_worker_model, args=(world, _free_port(), out, "fp32"), nprocs=world, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert torch.equal(r0["p"], r1["p"])
    ref_p, ref_losses = _single_process_reference()
    # Adam's first steps are sign-like: on the few hundred elements whose gradient is analytically zero (biases in front of a
    # GroupNorm: pure fp32 summation noise, other in the two-halves order than in the full batch) the update differs by up to lr per
    # step; everywhere else the two trajectories agree to rounding
    d = (r0["p"] - ref_p).abs()
    assert float((d > 2e-7 + 2e-5 * ref_p.abs()).float().mean()) < 1e-3, float((d > 2e-7 + 2e-5 * ref_p.abs()).float().mean())
    assert float(d.max()) <= 2.1 * 2 * 2e-4, float(d.max())
    assert float((r0["p"] - ref_p).norm() / ref_p.norm()) < 1e-5
    for a, b in zip(r0["losses"], ref_losses):          # the logged loss = mean over the ranks of the half-batch means
        assert abs(a - b) < 1e-5 * abs(b)


def test_bf16_gradient_wire_format(tmp_path):
    """JG_GRAD_WIRE=bf16 (parallel.GRAD_WIRE): the chunks travel as bf16 and are widened back into the fp32 arena before the optimizer;
    replicas stay bit-identical, and the step differs from the fp32 wire only by one bf16 rounding of the summed gradient (Adam turns a
    2^-9 relative gradient error into at most an lr-sized change of an element whose moment ratio sits at a rounding boundary)"""
    world = 2
    out = str(tmp_path / "bw%d.pt")
    try:
        mp.spawn(_worker_model, args=(world, _free_port(), out, "bf16"), nprocs=world, join=True)
    except Exception as e:       # a gloo build without bf16 all-reduce: the wire format is an RCCL-side option
        if "bfloat16" in str(e).lower() or "BFloat16" in str(e):
            import pytest
            pytest.skip("gloo in this torch build cannot all-reduce bf16")
        raise
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert torch.equal(r0["p"], r1["p"])
    ref_p, _ = _single_process_reference()
    d = (r0["p"] - ref_p).abs()
    assert float(d.max()) <= 2.1 * 2 * 2e-4                   # two steps of at most lr each way
    assert float((d > 1e-6).float().mean()) < 0.5             # most elements agree to rounding
    assert float(((r0["p"] - ref_p).norm()) / (ref_p.norm())) < 1e-4


# ---- the CUT step (three arenas: G, F, D) over two ranks, model level ------------------------------------------------------------------
CUT_CFG = dict(ngf=8, n_blocks=2, ndf=8, S=32, B=2, nce_layers=[0, 4, 8, 10, 11], num_patches=32)


def _cut_inputs():
    from joligen_amd.modules.cut_networks import PatchSampleF
    from joligen_amd.modules.discriminators import NLayerDiscriminator
    from joligen_amd.modules.resnet_generator import ResnetGenerator

    c = CUT_CFG
    torch.manual_seed(0)
    netG = ResnetGenerator(3, 3, c["ngf"], n_blocks=c["n_blocks"])
    netF = PatchSampleF(use_mlp=True, nc=32)
    netF.data_dependent_initialize(None, netG.feat_channels(c["nce_layers"]))
    netD = NLayerDiscriminator(3, c["ndf"])
    sdG = O.synth_state_dict({k: v.detach() for k, v in netG.state_dict().items()}, 0)
    sdF = O.synth_state_dict({k: v.detach().float() for k, v in netF.state_dict().items()}, 3)
    sdD = O.synth_state_dict({k: v.detach().float() for k, v in netD.state_dict().items()}, 1)
    g = torch.Generator().manual_seed(33)
    steps = []
    for _ in range(2):
        A = torch.rand(c["B"], 3, c["S"], c["S"], generator=g) * 2 - 1
        Bi = torch.rand(c["B"], 3, c["S"], c["S"], generator=g) * 2 - 1
        steps.append((A, Bi))
    return sdG, sdF, sdD, steps


def _cut_trainer(sdG, sdF, sdD):
    import random

    c = CUT_CFG
    return O.OracleCUTTrainer(sdG, sdF, sdD, c["n_blocks"], c["nce_layers"], num_patches=c["num_patches"], T=0.07, monce=False, pool_size=0,
                              pool_rng=random.Random(0), ema_beta=None)


def _cut_ids(tr, A, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        hw = [f.shape[2] * f.shape[3] for f in tr._feats(tr.G, A[:1])]
    return [[torch.randperm(n, generator=g)[:min(CUT_CFG["num_patches"], n)] for n in hw] for _ in range(2)]


class _DictArena(FakeArena):
    """flat view of one parameter dict of the OracleCUTTrainer (G, F or D)"""

    def __init__(self, params):
        self.names = list(params.keys())
        self.shapes = [params[k].shape for k in self.names]
        self.p = torch.cat([params[k].reshape(-1) for k in self.names])
        self.numel = self.p.numel()
        self.g, self.m, self.v = torch.zeros(self.numel), torch.zeros(self.numel), torch.zeros(self.numel)
        self.ema, self.step, self.dirty, self.calls = None, 0, False, []

    def load_grads(self, grads):
        self.g += torch.cat([grads[k].reshape(-1) for k in self.names])

    def store(self, params):
        off = 0
        for k, s in zip(self.names, self.shapes):
            n = params[k].numel()
            params[k] = self.p[off:off + n].reshape(s).clone()
            off += n


def _worker_cut(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from joligen_amd import parallel

    sdG, sdF, sdD, steps = _cut_inputs()
    tr = _cut_trainer(sdG, sdF, sdD)
    arenas = {"G": _DictArena(tr.G), "F": _DictArena(tr.Fp), "D": _DictArena(tr.D)}
    stores = {"G": tr.G, "F": tr.Fp, "D": tr.D}
    for a in arenas.values():
        parallel.broadcast_params(a, 0)
    half = CUT_CFG["B"] // world
    sl = slice(rank * half, (rank + 1) * half)
    hpG = dict(lr=2e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False, ema_beta=None, zero_grad=True)
    hpD = dict(hpG, lr=1e-4)
    for it, (A, Bi) in enumerate(steps):
        ids = _cut_ids(tr, A, 70 + it)                     # the same patch ids on every rank (PatchSampleF shares them over the batch)
        tr.iteration(A[sl], Bi[sl], ids[0], ids[1], iter_size=10 ** 9)      # gradients of this rank's half, no local optimizer step
        for name, hp in (("G", hpG), ("F", hpG), ("D", hpD)):               # group G's two arenas, then group D's (base_model.py:1302-1377)
            arenas[name].load_grads(tr.last_grads[name])
            arenas[name].step += 1
            parallel.allreduce_and_step(arenas[name], hp, grad_scale=1.0, n_chunks=4)
            arenas[name].store(stores[name])
    torch.save({k: a.p for k, a in arenas.items()}, out % rank)
    dist.destroy_process_group()


def test_cut_step_two_ranks_equals_single_process_batch(tmp_path):
    """the CUT optimisation step (resnet G + PatchGAN D + mlp_sample F, PatchNCE + lsgan; three gradient arenas stepped in the reference's
    group order) of a batch of 2 split over two ranks through parallel.allreduce_and_step == the single-process step on the whole batch:
    every term of the step is a mean over images of per-image quantities (InstanceNorm, within-image negatives, patch-logit means), so the
    mean of the per-rank gradients is the full-batch gradient (DDP semantics, models/base_model.py:725-737)."""
    world = 2
    out = str(tmp_path / "cut%d.pt")
    mp.spawn(_worker_cut, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    sdG, sdF, sdD, steps = _cut_inputs()
    tr = _cut_trainer(sdG, sdF, sdD)
    tr.hp.update(lr_G=2e-4, lr_D=1e-4)
    for it, (A, Bi) in enumerate(steps):
        ids = _cut_ids(tr, A, 70 + it)
        tr.step(A, Bi, ids[0], ids[1])
    for name, params, lr in (("G", tr.G, 2e-4), ("F", tr.Fp, 2e-4), ("D", tr.D, 1e-4)):
        assert torch.equal(r0[name], r1[name]), name
        ref = torch.cat([params[k].reshape(-1) for k in params])
        d = (r0[name] - ref).abs()
        # sign-like first Adam steps on analytically-zero gradients (conv biases in front of InstanceNorm) differ by up to lr per step
        assert float((d > 2e-7 + 2e-5 * ref.abs()).float().mean()) < 0.05, (name, float((d > 2e-7 + 2e-5 * ref.abs()).float().mean()))
        assert float(d.max()) <= 2.1 * 2 * lr, (name, float(d.max()))
        assert float((r0[name] - ref).norm() / ref.norm()) < 5e-4, (name, float((r0[name] - ref).norm() / ref.norm()))   # those few elements: <= 2 lr each

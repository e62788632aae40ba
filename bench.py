"""bench.py -- train images/sec of the DDPM (palette_model) step on N MI355X GPUs of one node.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU,
RCCL).  W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize; the MAX
over ranks is reported; rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): palette_model DDPM, efficient UNet (ngf 64, mults 1-2-4-8,
2 res-blocks/level, mid-block attention 16 heads x 32), 256x256, batch 32 PER GPU (weak scaling),
inpainting task with synthetic rectangle masks, AdamW + EMA (the example JSON's train section),
iter_size 1.  = examples/example_ddpm_noglasses2glasses.json + the overrides of SURVEY.md
Appendix C (data_crop_size 256, train_batch_size 32, train_iter_size 1,
G_unet_mha_vit_efficient true).  A step = set_input(device-resident batch) + optimize_parameters()
(forward, backward, gradient all-reduce, fused AdamW + EMA, refresh of the bf16 weights).

Default run: 50 timed steps; `ms_per_step` / `value` follow the contract (K steps / total time, max over ranks),
`ms_per_step_median` is the median of the per-step HIP-event times of the same region.  With the default flags (N = 1, palette) the
line also carries a `cut` object: the CUT G+D step of BASELINE configs[2] (SegFormer-attn G + [projected_d, basic] D + MoNCE,
256x256, batch 16) measured in the same process right after the palette leg, with its own roofline and CPU baseline, and the objects
`c4_512` (configs[3]: DDPM UNet 512x512, batch 8) and `cm` (configs[4]: consistency-model step 256x256, batch 64): value, ms per step,
fraction of the MFMA peak.

Extra objects on the JSON line:
  roofline     -- dominant kernel = the one with the largest summed time (conv3x3_halo_kernel: halo-resident
                  implicit-GEMM 3x3 convolution, forward + input-gradient):
                  achieved = algorithmic FLOPs per launch (2*M*N*K) / average launch duration, both
                  from HIP events recorded around EVERY launch on the launch stream during a
                  separate instrumented pass of 2 steps (the timed region carries no events) with the
                  weight-gradient side stream off, i.e. one kernel on the GPU at a time (what rocprofv3's
                  PMC passes see; `avg_launch_us_overlapped` = the same launches with the side stream on);
                  peak = 2500 TFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md); traffic = HBM bytes per launch
                  from the committed rocprofv3 PMC passes of the same command (profiles/r02_pmc.json).
  cpu_baseline -- the CPU oracle (oracle/jg_oracle.py, a torch-CPU-fp32 port of the reference
                  step) timed on this box's host cores on a bounded sample (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime initialises: see joligen_amd/__init__.py
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # multi-process GPU work on this stack needs dmabuf IPC (RCCL); exported on the boxes, kept here for ranks launched by hand

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0
# DDPM UNet, in-ch 6, forward GFLOP per image (conv/linear/attention matmuls x2)
FWD_GFLOP_PER_IMG = {(256, True): 369.78, (256, False): 413.27, (128, True): 92.04, (128, False): 102.91,
                     (512, True): 1504.88, (512, False): 1678.83}


def synth_batch(B, S, seed, device):
    """A,B ~ U(-1,1); one rectangle per image covering 10-40 % of the area;
    A = B (1-m) + N(0,1) m  (mirrors fill_mask_with_random)."""
    g = torch.Generator().manual_seed(seed)
    Bimg = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    m = torch.zeros(B, 1, S, S, dtype=torch.int64)
    for i in range(B):
        frac = float(torch.rand(1, generator=g)) * 0.3 + 0.1
        ar = float(torch.rand(1, generator=g)) * 1.0 + 0.5
        hh = max(1, min(S, int(round((frac * S * S * ar) ** 0.5))))
        ww = max(1, min(S, int(round(frac * S * S / hh))))
        h0 = int(torch.randint(0, S - hh + 1, (1,), generator=g))
        w0 = int(torch.randint(0, S - ww + 1, (1,), generator=g))
        m[i, :, h0:h0 + hh, w0:w0 + ww] = 1
    A = Bimg * (1 - m) + torch.randn(B, 3, S, S, generator=g) * m
    return {"A": A.to(device), "B": Bimg.to(device), "B_label_mask": m.to(device), "A_img_paths": ["synthetic"] * B}


def build_model(args, rank, local_rank, world):
    from joligen_amd.models import create_model
    from joligen_amd.options import opt_from_json

    ov = dict(model_type=args.model, G_netG="unet_mha", G_ngf=64, G_unet_mha_channel_mults=[1, 2, 4, 8],
              G_unet_mha_res_blocks=[2, 2, 2, 2], G_unet_mha_attn_res=[16], G_unet_mha_num_head_channels=32,
              G_unet_mha_vit_efficient=bool(args.efficient), data_crop_size=args.size, data_load_size=args.size,
              train_batch_size=args.batch, train_iter_size=1, train_optim="adamw", train_G_ema=True,
              train_G_ema_beta=0.999, train_G_lr=2e-4, alg_diffusion_task="inpainting", alg_palette_loss="MSE",
              gpu_ids=",".join(str(i) for i in range(world)), jg_act_dtype=args.dtype, name="bench",
              checkpoints_dir="/tmp/jg_bench_ckpt/")
    if args.model == "cut":
        # cut_model: resnet_9blocks / SegFormer-attn G + basic PatchGAN D + mlp_sample F, MoNCE, nce_idt, lsgan (example_gan_*.json shape)
        ov = dict(model_type="cut", G_netG=args.netG, G_ngf=64, G_nblocks=9, D_netDs=args.netDs.split(","), D_ndf=64, D_proj_interp=args.size,
                  D_proj_network_type=getattr(args, "proj", "efficientnet"), data_crop_size=args.size,
                  data_load_size=args.size, train_batch_size=args.batch, train_iter_size=1, train_optim="adam", train_G_ema=True,
                  train_G_ema_beta=0.999, gpu_ids=",".join(str(i) for i in range(world)), jg_act_dtype=args.dtype, name="bench",
                  checkpoints_dir="/tmp/jg_bench_ckpt/")
    opt = opt_from_json({}, ov)
    torch.manual_seed(0)  # reference-style default init (+ zero_module); identical on every rank
    model = create_model(opt, local_rank if world > 1 else 0)
    if args.model == "cut":          # train.py:197-199: the feature network is shaped by a first batch
        g0 = torch.Generator().manual_seed(1)
        dev0 = torch.device("cuda", local_rank)
        model.data_dependent_initialize({"A": (torch.rand(args.batch, 3, args.size, args.size, generator=g0) * 2 - 1).to(dev0),
                                         "B": (torch.rand(args.batch, 3, args.size, args.size, generator=g0) * 2 - 1).to(dev0)})
    model.setup(opt)
    if world > 1 or args.force_exchange:
        model.parallelize(local_rank)
    else:
        model.single_gpu()
    return model, opt


def usable_cores():
    """Host cores this process may actually run on: min(affinity mask, cgroup CPU quota).
    (os.cpu_count() reports the whole machine; oversubscribing a quota-limited container with one
    OpenMP thread per machine core makes the CPU run orders of magnitude slower.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def _reference_ratio(model):
    """the port-vs-reference ratio measured once in the build container (oracle/measure_reference_cpu.py; /root/reference does not
    exist on the GPU box, so the timed CPU leg is the oracle PORT of the reference step): palette (round 2), cut / cm (round 6, the
    configs[2] / configs[4] shapes the legs time).  Returns (note for `sample`, port speed / reference speed)."""
    try:
        if model == "palette":
            r = json.load(open(os.path.join(ROOT, "profiles", "r02_cpu_reference_vs_port.json")))
        else:
            r = json.load(open(os.path.join(ROOT, "profiles", "r06_cpu_reference_vs_port.json")))[model]
        return (f"; kind=port: the unmodified reference itself measured {r['reference_img_per_s']} images/s against {r['port_img_per_s']} for this "
                f"port on the build container's {r['cores']} cores (port / reference = {r['port_over_reference']}x)"), r["port_over_reference"]
    except Exception:
        return "", None


def cpu_baseline_subprocess(args, timeout_s=240):
    """Run the CPU leg in a child process with a hard wall-clock bound so that bench.py always
    finishes within minutes whatever the host looks like."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(args.size),
           "--efficient", str(args.efficient), "--model", args.model, "--netG", args.netG, "--netDs", args.netDs, "--proj", getattr(args, "proj", "efficientnet")]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for line in out.stdout.splitlines():
            if line.startswith("{"):
                res = json.loads(line)
                note, ratio = _reference_ratio(args.model)
                res["sample"] += note
                res["port_over_reference"] = ratio
                return res
        return {"value": None, "unit": "images/sec", "cores": usable_cores(), "kind": "port",
                "sample": "cpu leg failed: " + out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "images/sec", "cores": usable_cores(), "kind": "port",
                "sample": f"cpu leg exceeded its {timeout_s}s bound on this host"}


def cpu_baseline(args):
    """The oracle's full step (forward, backward, AdamW, EMA) on the host cores, bounded sample."""
    import jg_oracle as O
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.options import opt_from_json

    cores = usable_cores()
    torch.set_num_threads(cores)
    S, Bc = args.size, 1
    opt = opt_from_json({}, dict(G_unet_mha_vit_efficient=bool(args.efficient), data_crop_size=S, model_type=args.model))
    torch.manual_seed(0)
    batch = synth_batch(Bc, S, 99, "cpu")
    gen = torch.Generator().manual_seed(3)
    if args.model == "cut":
        import random

        from joligen_amd.modules.discriminators import NLayerDiscriminator
        from joligen_amd.modules.cut_networks import PatchSampleF
        seg = "segformer" in args.netG
        if seg:
            from joligen_amd.modules.segformer import SegformerGenerator_attn
            netG = SegformerGenerator_attn(None, None, 3, S, 10, 1)
            layers, T = [0, 1, 2, 3], 0.2
        elif "resnet_attn" in args.netG:
            from joligen_amd.modules.resnet_attn_generator import ResnetGenerator_attn
            netG = ResnetGenerator_attn(3, 3, 10, 1, 64, n_blocks=9, mobile=args.netG.startswith("mobile"))
            layers, T = [0, 4, 8, 12, 16], 0.07
        else:
            from joligen_amd.modules.resnet_generator import ResnetGenerator
            netG = ResnetGenerator(3, 3, 64, n_blocks=9)
            layers, T = [0, 4, 8, 12, 16], 0.07
        netF = PatchSampleF(use_mlp=True)
        netF.data_dependent_initialize(None, netG.feat_channels(layers))
        sdG = {k: v.detach() for k, v in netG.state_dict().items()}
        sdF = {k: v.detach().float() for k, v in netF.state_dict().items()}
        sdD = {k: v.detach().float() for k, v in NLayerDiscriminator(3, 64).state_dict().items()}
        sdPD = None
        if "projected_d" in args.netDs.split(","):
            import warnings

            from joligen_amd.modules.projected_d import ProjectedDiscriminator
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                sdPD = {k: v.detach() for k, v in ProjectedDiscriminator(getattr(args, "proj", "efficientnet"), interp=S, img_size=S).state_dict().items()}
        tr = O.OracleCUTTrainer(sdG, sdF, sdD, 9, layers, num_patches=256, T=T, monce=True, pool_size=50, pool_rng=random.Random(0),
                                ema_beta=0.999, gen="segformer" if seg else (args.netG if "resnet_attn" in args.netG else "resnet"),
                                sdPD=sdPD, proj_interp=S if sdPD is not None else -1)
        A, Bm = batch["A"], batch["B"]
        with torch.no_grad():
            hw = [f.shape[2] * f.shape[3] for f in (O.segformer_backbone(sdG, A) if seg else tr._feats(sdG, A))]
        times = []
        for it in range(9):
            ids = [[torch.randperm(n, generator=gen)[:min(256, n)] for n in hw] for _ in range(2)]
            uni = None
            if seg:   # 14 DropPath draws + 2 Dropout2d draws [., 256] for the generator forward, then 4 x 14 DropPath draws
                uni = [torch.rand(2 * Bc, generator=gen) for _ in range(14)] + [torch.rand(2 * Bc, 256, generator=gen) for _ in range(2)] \
                    + [torch.rand(Bc, generator=gen) for _ in range(56)]      # the generator runs on cat(A, B); the 4 encoder passes on B images
            t0 = time.perf_counter()
            tr.step(A, Bm, ids[0], ids[1], uniforms=uni)
            times.append(time.perf_counter() - t0)
            if it and sum(times[1:]) > 15.0:
                break
        per_step = sum(times[1:]) / len(times[1:])
        return {"value": round(Bc / per_step, 4), "unit": "images/sec", "cores": cores, "kind": "port",
                "sample": f"oracle/jg_oracle.py OracleCUTTrainer ({args.netG} G + D_netDs [{args.netDs}] + F, MoNCE), {len(times) - 1} timed full iterations "
                          f"(G/F group + D group, {3 + (sdPD is not None)} Adam steps, EMA) of batch {Bc} at {S}x{S} fp32 after 1 warm-up, {cores} torch threads"}
    if args.model == "cm":
        from joligen_amd.models.cm_model import define_G_cm

        opt.alg_diffusion_cond_embed_dim = 256
        sd = {k: v.detach().float() for k, v in define_G_cm(opt).state_dict().items()}
        tr = O.OracleCMTrainer(sd, O.UNetCfg(in_channel=3, efficient=bool(args.efficient), cond_embed_dim=256), 1000000)
        times = []
        for it in range(17):
            noise, ts = O.cm_draw_step_randomness(gen, batch["B"], tr.sigmas())
            t0 = time.perf_counter()
            tr.optimize_parameters(batch["B"], batch["B_label_mask"], noise, ts)
            times.append(time.perf_counter() - t0)
            if it and sum(times[1:]) > 15.0:
                break
        per_step = sum(times[1:]) / len(times[1:])
        return {"value": round(Bc / per_step, 4), "unit": "images/sec", "cores": cores, "kind": "port",
                "sample": f"oracle/jg_oracle.py OracleCMTrainer, {len(times) - 1} timed full steps (2 fwd + bwd + AdamW + EMA) "
                          f"of batch {Bc} at {S}x{S} fp32 after 1 warm-up, {cores} torch threads"}
    sd = {k: v.detach().float() for k, v in define_G(**vars(opt)).state_dict().items()}
    cfg = O.UNetCfg(efficient=bool(args.efficient))
    tr = O.OraclePaletteTrainer(sd, cfg)
    times = []
    budget_s, t_start = 15.0, time.perf_counter()   # bounded sample: ~15 s of CPU work after the warm-up call
    for it in range(33):  # first call is the warm-up
        t, u, noise = O.draw_step_randomness(gen, batch["B"], 2000)
        t0 = time.perf_counter()
        tr.optimize_parameters(batch["B"], batch["A"], batch["B_label_mask"], noise, t, u)
        times.append(time.perf_counter() - t0)
        if it == 0:
            t_start = time.perf_counter()
        elif time.perf_counter() - t_start > budget_s:
            break
    per_step = sum(times[1:]) / len(times[1:])
    return {"value": round(Bc / per_step, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/jg_oracle.py OraclePaletteTrainer, {len(times) - 1} timed full steps (fwd+bwd+AdamW+EMA) "
                      f"of batch {Bc} at {S}x{S} fp32 after 1 warm-up, {cores} torch threads"}


HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
RIDGE_FLOP_PER_BYTE = PEAK_BF16_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)


def csrc_sha():
    """sha1 of the kernel sources (the same digest tools/collect_evidence.sh stamps into the committed profiles)"""
    import glob
    import hashlib

    h = hashlib.sha1()
    d = os.path.join(ROOT, "joligen_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip"))) + sorted(glob.glob(os.path.join(d, "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def timed_region(step, steps, warmup, fence):
    """W untimed steps, then exactly K steps between two fences; returns (seconds of the K steps, per-step milliseconds from HIP events
    recorded on the compute stream after every step -- two event records per step, no synchronisation inside the region)"""
    for _ in range(warmup):
        step()
    fence()
    mark = os.environ.get("JG_TRACE_MARK", "0") != "0"      # profiling runs: a uniquely named kernel (at::cuda's spin_kernel) brackets the
    if mark:                                                 # timed steps, tools/rocpd_stats.py then counts the steady-state window only
        torch.cuda._sleep(2000)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step()
        evs[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    if mark:
        torch.cuda._sleep(2000)
        torch.cuda.synchronize()
    return dt, [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]


def kernel_table(recs, nsteps):
    """per kernel INSTANCE (the name the dispatch recorded, jg_last_kernel): launches, time, algorithmic FLOPs and HBM bytes; each row
    is priced against the roof its arithmetic intensity puts it under (ridge = 2500 TFLOP/s / 8 TB/s = 312 FLOP/B)"""
    per = {}
    for name, e0, e1, fl, _geo, by in recs:
        d = per.setdefault(name, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += fl
        d[3] += by
    out = {}
    for k, (n, t, f, b) in per.items():
        tf, gbs = f / t / 1e12, b / t / 1e9
        mfma = f / max(b, 1.0) >= RIDGE_FLOP_PER_BYTE
        out[k] = {"bound": "mfma" if mfma else "hbm", "achieved": round(tf if mfma else gbs, 2), "peak": PEAK_BF16_TFLOPS if mfma else HBM_PEAK_GBS,
                  "unit": "TFLOP/s" if mfma else "GB/s", "frac": round((tf / PEAK_BF16_TFLOPS) if mfma else (gbs / HBM_PEAK_GBS), 4),
                  "tflops": round(tf, 2), "algorithmic_gbs": round(gbs, 1), "launches_per_step": n // nsteps, "avg_launch_us": round(t / n * 1e6, 2),
                  "avg_flops_per_launch": round(f / n, 1), "avg_bytes_per_launch": round(b / n, 1), "time_per_step_ms": round(t / nsteps * 1e3, 3)}
    return out


def cut_leg(local_rank, no_cpu, proj="vitsmall"):
    """BASELINE configs[2] on one GPU: cut_model, SegFormer-attn G + [projected_d, basic] D + mlp_sample F + MoNCE, 256x256, batch 16, bf16.
    `proj` = D_proj_network_type: "vitsmall" (what examples/example_gan_mario2sonic.json selects: vit_small_patch16_224 at proj_interp 256, the
    `cut` object of the JSON line, round 5) or "efficientnet" (tf_efficientnet_lite0, the `cut_effnet` object = the `cut` selection of rounds
    3 - 4, kept for continuity); both with random frozen weights (no timm checkpoint offline)."""
    import warnings

    from joligen_amd import ops

    ns = argparse.Namespace(model="cut", netG="segformer_attn_conv", netDs="projected_d,basic", batch=16, size=256, dtype="bf16", efficient=1,
                            force_exchange=False, proj=proj)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, _ = build_model(ns, 0, local_rank, 1)
    dev = torch.device("cuda", local_rank)
    g = torch.Generator().manual_seed(77)
    batch = {"A": (torch.rand(ns.batch, 3, ns.size, ns.size, generator=g) * 2 - 1).to(dev), "B": (torch.rand(ns.batch, 3, ns.size, ns.size, generator=g) * 2 - 1).to(dev)}

    def step():
        model.set_input(batch)
        model.optimize_parameters()

    steps, warmup = 20, 5
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        dt, per_step = timed_region(step, steps, warmup, torch.cuda.synchronize)
    import joligen_amd
    # which driver ran the timed steps: "graph+graphG" = the generator half (forward graph, backward graph) AND the
    # discriminator half replayed from hipGraphs, "graph" = the discriminator half only, "early" = that half eager on a second stream,
    # "sequential" = the reference's order; the canary's verdict and any jg_graph_D warning travel with the number
    driver = {"step_driver": model.step_driver, "hip_graphs_safe": bool(joligen_amd.HIP_GRAPHS_SAFE),
              "graph_canary": ("passed" if model.step_driver == "graph+graphG" else ("failed" if "graph dropped" in model.step_driver_note else
                               "discriminator half only" if model.step_driver == "graph" else "not run")),
              "note": model.step_driver_note, "warnings": [str(w.message)[:300] for w in rec if "jg_" in str(w.message)][:4],
              "DEBUG_CLR_GRAPH_PACKET_CAPTURE": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")}
    for w in rec:
        print(f"[cut leg] warning: {w.message}", file=sys.stderr, flush=True)
    print(f"[cut leg] step_driver={driver['step_driver']} canary={driver['graph_canary']} {driver['note']}", file=sys.stderr, flush=True)
    ops.KERNEL_TIMING = []
    step()
    torch.cuda.synchronize()
    recs, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
    table = kernel_table(recs, 1)
    dom = max(table, key=lambda k: table[k]["time_per_step_ms"])
    # HBM traffic of the dominant instance: the committed rocprofv3 PMC passes of this step (profiles/r05_cut_pmc.json, tools/collect_evidence.sh cutpmc: a filtered PMC pass over the conv / weight-gradient family)
    traffic, traffic_build, traffic_file = None, None, None
    try:
        fn = next(f for f in ("r06_cut_pmc.json", "r05_cut_pmc.json", "r04_cut_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
        pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
        row = pmc.get(dom) or pmc.get(dom.split("<")[0])
        if row:
            traffic, traffic_build, traffic_file = round(row["bytes_per_launch"], 1), pmc.get("_meta", {}).get("build"), "profiles/" + fn
    except Exception:
        pass
    roof = dict(table[dom], kernel=dom, traffic=traffic, traffic_unit="HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
                traffic_source=traffic_file, traffic_build=traffic_build, running_build="csrc sha1 " + csrc_sha(),
                measured="HIP events around every convolution-family launch of one instrumented step (the step is ~4000 launches of 5 - 30 us: "
                         "small-problem latency next to a few HBM-bound streaming kernels, DESIGN.md 11); achieved = algorithmic bytes (or FLOPs) "
                         "of the dominant instance / its summed launch time (events around a 10 us launch include ~5 us of dispatch)",
                conv_family_ms_per_step=round(sum(r["time_per_step_ms"] for r in table.values()), 3),
                other_kernels={k: v for k, v in table.items() if k != dom})
    cpu = None if no_cpu else cpu_baseline_subprocess(ns, timeout_s=180)
    loss = float(model.get_current_losses()["G_tot"].detach())
    return {"metric": "train images/sec at 256x256 (CUT G+D step)", "value": round(ns.batch * steps / dt, 3), "unit": "images/sec",
            "ms_per_step": round(dt / steps * 1e3, 3), "ms_per_step_median": round(sorted(per_step)[len(per_step) // 2], 3), "steps": steps, "warmup": warmup,
            "dtype": "bf16", "data": "synthetic", **driver,
            "config": {"workload": "cut_model, segformer_attn_conv G (MiT-b0 + heads + ResnetDecoder tail) + D_netDs [projected_d ("
                                   + ("vit_small_patch16_224 at proj_interp 256: 257 tokens, Conv1d CCM / vector CSM / 4 MLP heads" if proj == "vitsmall" else "tf_efficientnet_lite0")
                                   + " architecture, random frozen weights: timm checkpoint unavailable), basic] + mlp_sample F, MoNCE, nce_idt, hinge / lsgan, "
                                   "256x256, batch 16/GPU, Adam x4 + EMA, iter_size 1 (BASELINE configs[2] shape; example_gan_mario2sonic.json without "
                                   "vision_aided / semantic mask)", "global_batch": ns.batch, "final_loss": round(loss, 5)},
            "roofline": roof, "cpu_baseline": cpu}


def unet_leg(local_rank, model_kind, size, batch, efficient, steps=20, warmup=5, no_cpu=True):
    """One more single-GPU configuration of BASELINE.json on the default line: `c4_512` = configs[3] (palette_model DDPM, UNet with mid-block
    self-attention, 512x512, batch 8 per GPU) and `cm` = configs[4] (cm_model consistency step, 256x256, batch 64 per GPU, fused AdamW):
    value, ms per step, the step's algorithmic FLOPs as a fraction of the bf16 MFMA peak.  Same step definition as the palette leg
    (set_input on a device-resident batch + optimize_parameters()); run by `leg_subprocess` in a process of its own."""
    ns = argparse.Namespace(model=model_kind, netG="resnet", netDs="basic", batch=batch, size=size, dtype="bf16", efficient=int(efficient),
                            force_exchange=False)
    model, _ = build_model(ns, 0, local_rank, 1)
    data = synth_batch(batch, size, 4321, torch.device("cuda", local_rank))

    def step():
        model.set_input(data)
        model.optimize_parameters()

    dt, per_step = timed_region(step, steps, warmup, torch.cuda.synchronize)
    ms = dt / steps * 1e3
    mult = 3 if model_kind == "palette" else 4          # cm = student forward + teacher forward + backward
    tflop = mult * FWD_GFLOP_PER_IMG[(size, bool(efficient))] * batch / 1e3
    loss = float(model.get_current_losses()["G_tot"].detach())
    cpu = None if no_cpu else cpu_baseline_subprocess(ns, timeout_s=180)
    return {"cpu_baseline": cpu, "metric": f"train images/sec at {size}x{size} ({'DDPM UNet' if model_kind == 'palette' else 'CM UNet'} step)",
            "value": round(batch * steps / dt, 3), "unit": "images/sec", "ms_per_step": round(ms, 3),
            "ms_per_step_median": round(sorted(per_step)[len(per_step) // 2], 3), "steps": steps, "warmup": warmup, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{'palette_model DDPM' if model_kind == 'palette' else 'cm_model consistency'}, {'efficient ' if efficient else ''}UNet unet_mha "
                                   f"ngf64 mults[1,2,4,8] res_blocks[2,2,2,2] mid-attn 16x32, {size}x{size}, batch {batch}/GPU, inpainting synthetic masks, "
                                   "AdamW+EMA, iter_size 1", "global_batch": batch, "final_loss": round(loss, 6)},
            "step_algorithmic_tflop": round(tflop, 3), "step_frac_of_mfma_peak": round(tflop / (ms * 1e-3) / PEAK_BF16_TFLOPS, 4)}


# BASELINE configs[3] and configs[4] at their own shapes
EXTRA_LEGS = {"c4_512": dict(model_kind="palette", size=512, batch=8, efficient=False), "cm": dict(model_kind="cm", size=256, batch=64, efficient=True)}


ROOF_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_per_step", "avg_launch_us", "time_per_step_ms",
             "avg_flops_per_launch", "avg_bytes_per_launch", "traffic_source", "traffic_build", "running_build", "conv3x3_family",
             "avg_launch_us_overlapped", "step_algorithmic_tflop", "step_frac_of_mfma_peak", "conv_family_ms_per_step")


def compact_roofline(r):
    return None if not r else {k: r[k] for k in ROOF_KEEP if k in r}


def compact_cpu(c):
    if not c:
        return c
    out = {k: c[k] for k in ("value", "unit", "cores", "kind") if k in c}
    out["sample"] = str(c.get("sample", ""))[:200]
    if "port_over_reference" in c:
        out["port_over_reference"] = c["port_over_reference"]
    return out


def compact_leg(leg):
    """An extra leg (cut / cut_effnet / c4_512 / cm) as the few numbers a reader of the record needs; the full object is in the detail line."""
    if not isinstance(leg, dict):
        return leg
    # ms_per_step_median next to the mean: the pool's boxes are shared (another tenant's job showed 183 GB of VRAM in use on an "idle" box and one
    # cm leg in ~20 read 327 ms per step against 121): the median of the same K steps says whether a mean is a stall or the step
    out = {k: leg[k] for k in ("value", "unit", "ms_per_step", "ms_per_step_median", "steps", "dtype", "step_driver", "graph_canary", "step_frac_of_mfma_peak", "error")
           if k in leg}
    r = leg.get("roofline")
    if r:
        out["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_us")}
    c = leg.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "port_over_reference")}
    w = (leg.get("config") or {}).get("workload")
    if w:
        out["config"] = {"workload": w[:120]}
    return out


def emit(line):
    """The record keeps the TAIL of stdout and the last line must be the bench line: the full objects (per-kernel tables of every leg, ~20 KB in
    round 5, which pushed the `cut` value out of the driver's record) go out FIRST as a `bench_detail` line (and to gpurun_out/ when that exists);
    the LAST line is the same measurement under 4 KB: the palette line with its dominant-kernel roofline and CPU baseline, every leg as a
    compact object."""
    detail = json.dumps({"bench_detail": line})
    print(detail, flush=True)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                f.write(json.dumps(line, indent=1))
    except OSError:
        pass
    short = dict(line)
    short["roofline"] = compact_roofline(line.get("roofline"))
    short["cpu_baseline"] = compact_cpu(line.get("cpu_baseline"))
    if isinstance(short.get("config"), dict):
        short["config"] = dict(short["config"], workload=short["config"].get("workload", "")[:260])
    for k in ("cut", "cut_effnet", "c4_512", "cm"):
        if k in short:
            short[k] = compact_leg(short[k])
    short["detail"] = "full per-kernel tables of every leg: the preceding `bench_detail` stdout line (gpurun_out/bench_detail.json on the GPU box)"
    out = json.dumps(short)
    if len(out) > 4096:      # never let the tables back in by accident
        for k in ("cut", "cut_effnet", "c4_512", "cm"):
            if isinstance(short.get(k), dict):
                short[k].pop("config", None)
        out = json.dumps(short)
    print(out, flush=True)


def leg_subprocess(name, no_cpu, timeout_s=420):
    """One extra leg of the default line in its OWN process: every leg starts from an empty allocator (in one process the legs inherited
    each other's 100+ GB of cached blocks, and one step in ten of the 512x512 leg stalled for ~2 s in hipMalloc), and a leg that dies
    cannot take the palette line with it."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--leg", name] + (["--no-cpu-baseline"] if no_cpu else [])
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        if out.stderr.strip():              # the leg's own warnings (e.g. a dropped hipGraph) reach the caller's log
            print(f"[bench leg {name}] stderr tail:\n" + out.stderr[-1500:], file=sys.stderr, flush=True)
        for line in reversed(out.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "error": ("leg failed: " + out.stderr[-300:])}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"leg exceeded its {timeout_s}s bound"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]: 32)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--efficient", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--netG", default="resnet", help="--model cut only: resnet (BASELINE configs[0] generator, 9 blocks) | segformer_attn_conv (configs[2]) | resnet_attn | "
                         "mobile_resnet_attn (examples/example_gan_horse2zebra.json)")
    ap.add_argument("--netDs", default="basic", help="--model cut only: comma list out of basic, projected_d (BASELINE configs[2]: projected_d,basic)")
    ap.add_argument("--proj", default="efficientnet", choices=["efficientnet", "vitsmall"], help="--model cut only: D_proj_network_type of projected_d")
    ap.add_argument("--model", default="palette", choices=["palette", "cm", "cut"],
                    help="palette = BASELINE configs[1] (the bench line); cm = the consistency-model step of configs[4] (same UNet)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cut-leg", action="store_true", help="skip the CUT (BASELINE configs[2]) leg of the default line")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--force-exchange", action="store_true", help="dev: run the multi-GPU gradient exchange path on one GPU (1-rank RCCL group)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--leg", default="", help=argparse.SUPPRESS)      # child mode: run ONE extra leg of the default line (cut | c4_512 | cm), print its JSON object
    ap.add_argument("--dump-kernel-timing", default="", help="write the per-shape conv kernel timing table here")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    if args.leg:
        torch.cuda.set_device(0)
        obj = (cut_leg(0, args.no_cpu_baseline) if args.leg == "cut" else cut_leg(0, True, proj="efficientnet") if args.leg == "cut_effnet"
               else unet_leg(0, **EXTRA_LEGS[args.leg], no_cpu=args.no_cpu_baseline or args.leg != "cm"))
        print(json.dumps(obj), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: re-launch under torch.distributed.run (one rank per GPU); the ranks inherit stdout, so
        # rank 0's JSON line is this process's last line as well
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if args.force_exchange:     # dev: run the data-parallel exchange path (1-rank RCCL group) on a single GPU
        from joligen_amd import parallel
        parallel.FORCE_EXCHANGE = True

    model, opt = build_model(args, rank, local_rank, world)
    batch = synth_batch(args.batch, args.size, 1234 + rank, device)
    if args.model == "cut":
        batch = {"A": batch["A"], "B": batch["B"]}

    def step():
        model.set_input(batch)
        model.optimize_parameters()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from joligen_amd import parallel as jg_parallel

    if world > 1 or args.force_exchange:
        jg_parallel.TIMING = []
    for _ in range(args.warmup):          # warm-up outside timed_region(): the exchange diagnostics start with the timed steps
        step()
    fence()
    if jg_parallel.TIMING is not None:
        jg_parallel.TIMING.clear()
    dt, per_step_ms = timed_region(step, args.steps, 0, fence)
    dt_local = dt
    exch = None
    if jg_parallel.TIMING:
        ex_ms = sorted(a.elapsed_time(b) for a, b in jg_parallel.TIMING)
        exch = {"wait_plus_optimizer_ms_median": round(ex_ms[len(ex_ms) // 2], 3), "wait_plus_optimizer_ms_max": round(ex_ms[-1], 3),
                "note": "compute-stream time between the first chunk wait and the last optimizer chunk of allreduce_and_step (exposed part of the "
                        "gradient exchange + the chunked fused optimizer; the optimizer alone is ~0.4 ms at this size)", "wire": jg_parallel.GRAD_WIRE}
    jg_parallel.TIMING = None
    # logging path of the reference (train.py:293-301): the printed loss is the mean over the ranks
    loss = float(model.get_current_losses_reduced()["G_tot"].detach())
    n_ranks_seen = dist.get_world_size() if dist.is_initialized() else 1
    per_rank = None
    if world > 1:
        # diagnostics of a scaling run: every rank's own wall time for the K steps and its own median step / exchange time
        mine = torch.tensor([dt_local / args.steps * 1e3, sorted(per_step_ms)[len(per_step_ms) // 2],
                             exch["wait_plus_optimizer_ms_median"] if exch else 0.0], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(float(t[0]), 3) for t in allr], "ms_per_step_median": [round(float(t[1]), 3) for t in allr],
                    "exchange_wait_plus_optimizer_ms_median": [round(float(t[2]), 3) for t in allr]}
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms_per_step = dt / args.steps * 1e3
    ms_median = sorted(per_step_ms)[len(per_step_ms) // 2]
    value = args.batch * world * args.steps / dt

    # ---- dominant-kernel roofline: instrumented pass, HIP events around every conv launch ----
    roofline = None
    if not args.no_kernel_timing:
        from joligen_amd import ops

        from joligen_amd.modules import unet_exec

        def instrumented():
            ops.KERNEL_TIMING = []
            step()
            step()
            torch.cuda.synchronize()
            r, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
            return r

        # in the timed region the weight gradients run on a second stream NEXT TO the compute stream's kernels: a
        # launch then shares the CUs with another kernel and its event-to-event time is not the kernel's own.  The roofline numbers are
        # taken with that side stream off (one kernel on the GPU at a time, what rocprofv3's PMC passes also see); the overlapped
        # average of the dominant kernel is reported next to it.
        recs_overlapped = instrumented() if unet_exec.WGRAD_STREAM else None
        side_was, unet_exec.WGRAD_STREAM = unet_exec.WGRAD_STREAM, False
        recs = instrumented()
        unet_exec.WGRAD_STREAM = side_was
        if args.dump_kernel_timing:
            agg = {}
            for name, e0, e1, fl, geo, _by in recs:
                a = agg.setdefault((name, geo), [0, 0.0, fl])
                a[0] += 1
                a[1] += e0.elapsed_time(e1)
            with open(args.dump_kernel_timing, "w") as f:
                for (name, geo), (n_, t_, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write(f"{name:40s} {str(geo):48s} calls/step {n_ / 2:5.1f} ms/step {t_ / 2:8.3f} avg_us {t_ / n_ * 1e3:9.1f} TF {fl / (t_ / n_ * 1e-3) / 1e12:7.1f}\n")
        table = kernel_table(recs, 2)
        # the dominant kernel is ONE instance (the dispatch's own name: the persistent Cin = 64 kernel is not a halo-kernel row)
        dom = max(table, key=lambda k: table[k]["time_per_step_ms"])
        # HBM traffic per launch of the dominant kernel: rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md "HBM") of
        # this same command, committed under profiles/ with the sha of the kernel sources they were taken on.  PMC counters cannot be
        # read from inside the timed process: `traffic` is that committed measurement, `traffic_build` says which build it is from and
        # `running_build` which one produced every other number of this line.
        traffic, traffic_build, traffic_file = None, None, None
        for cand in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json"):
            try:
                if args.model != "palette":
                    break
                pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                row = pmc.get(dom) or pmc.get(dom.split("<")[0])
                if row:
                    traffic, traffic_build, traffic_file = round(row["bytes_per_launch"], 1), pmc.get("_meta", {}).get("build"), "profiles/" + cand
                    break
            except Exception:
                pass
        roofline = dict(table[dom], kernel=dom, traffic=traffic, traffic_unit="HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)",
                        traffic_source=traffic_file, traffic_build=traffic_build, running_build="csrc sha1 " + csrc_sha(),
                        measured="HIP events around every launch of an instrumented pass of 2 steps with the weight-gradient side stream OFF "
                                 "(one kernel on the GPU at a time); the timed region runs with it on",
                        other_kernels={k: v for k, v in table.items() if k != dom})
        # all instances of the halo-resident 3x3 forward / input-gradient family together (what rounds 1-2 reported as one row)
        fam = [v for k, v in table.items() if k.startswith("conv3x3_halo_kernel") or k.startswith("conv3x3_p64_kernel")]
        if fam:
            ft = sum(v["time_per_step_ms"] for v in fam) * 1e-3
            ff = sum(v["avg_flops_per_launch"] * v["launches_per_step"] for v in fam)
            roofline["conv3x3_family"] = {"instances": len(fam), "launches_per_step": sum(v["launches_per_step"] for v in fam),
                                          "time_per_step_ms": round(ft * 1e3, 3), "achieved_tflops": round(ff / ft / 1e12, 2),
                                          "frac_of_mfma_peak": round(ff / ft / 1e12 / PEAK_BF16_TFLOPS, 4)}
        if recs_overlapped:
            ov = [e0.elapsed_time(e1) for name, e0, e1, fl, _g, _b in recs_overlapped if name == dom]
            roofline["avg_launch_us_overlapped"] = round(sum(ov) / max(len(ov), 1) * 1e3, 2)
        gf = FWD_GFLOP_PER_IMG.get((args.size, bool(args.efficient)))
        if gf and args.model != "cut":
            # palette: forward + backward (2x) = 3x; cm: student forward + teacher forward + backward = 4x
            step_tflop = (3 if args.model == "palette" else 4) * gf * args.batch / 1e3
            roofline["step_algorithmic_tflop"] = round(step_tflop, 3)
            roofline["step_frac_of_mfma_peak"] = round(step_tflop / (ms_per_step * 1e-3) / PEAK_BF16_TFLOPS, 4)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_subprocess(args)
    cut, extra = None, {}
    if rank == 0 and world == 1 and args.model == "palette" and not args.no_cut_leg and not args.force_exchange:
        # the CUT half of BASELINE's metric ("DDPM UNet & CUT G+D step") and configs[3] / configs[4] on the same line, each in its own
        # process and never at the palette line's expense
        # the palette numbers are taken: hand the parent's cached device memory back before the legs allocate theirs (the pool's boxes are shared --
        # 183 GB of VRAM in use by another tenant on an "idle" box -- and a cm leg (batch 64) that has to evict its way to 100 GB read one step of
        # 15 s among twenty of 121 ms: mean 872 ms, median 121 ms)
        model = None
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        print("[bench] cut leg", file=sys.stderr, flush=True)
        cut = leg_subprocess("cut", args.no_cpu_baseline)
        for key in list(EXTRA_LEGS) + ["cut_effnet"]:
            print(f"[bench] {key} leg", file=sys.stderr, flush=True)
            extra[key] = leg_subprocess(key, args.no_cpu_baseline or key != "cm")      # CPU leg: cut (above) and cm; c4_512 / cut_effnet share their families' ratios

    if rank == 0:
        line = {
            "metric": f"train images/sec at {args.size}x{args.size} ({'DDPM UNet' if args.model == 'palette' else 'CM UNet' if args.model == 'cm' else 'CUT'} step)",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "ms_per_step_median": round(ms_median, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"cut_model, {args.netG} G (ngf 64, 9 blocks) + D_netDs [{args.netDs}] (projected_d: {args.proj} architecture, random frozen weights, DESIGN.md 14) + mlp_sample F, MoNCE, nce_idt, lsgan / hinge, "
                                    f"{args.size}x{args.size}, batch {args.batch}/GPU, Adam x3 + EMA, iter_size 1") if args.model == "cut" else
                                   f"{'palette_model DDPM' if args.model == 'palette' else 'cm_model consistency'}, {'efficient ' if args.efficient else ''}UNet unet_mha ngf64 mults[1,2,4,8] "
                                   f"res_blocks[2,2,2,2] mid-attn 16x32, {args.size}x{args.size}, batch {args.batch}/GPU, "
                                   "inpainting synthetic masks, AdamW+EMA, iter_size 1 "
                                   "(example_ddpm_noglasses2glasses.json + SURVEY Appendix C overrides)",
                       "global_batch": args.batch * world, "per_gpu_batch": args.batch, "image_size": args.size,
                       "efficient": bool(args.efficient), "parallelism": f"dp{world}", "final_loss": round(loss, 6),
                       "n_ranks_seen": n_ranks_seen, **({"step_driver": model.step_driver, "step_driver_note": model.step_driver_note} if args.model == "cut" else {})},
            "roofline": roofline, "cpu_baseline": cpu, "cut": cut,
        }
        line.update(extra)
        if per_rank is not None:
            line["per_rank"] = per_rank
        if exch is not None:
            line["exchange"] = exch
    # RCCL prints its banner through C stdio (block-buffered when stdout is a pipe, i.e. written at exit): every rank flushes it
    # now, so that rank 0's JSON line is the LAST line of the job's stdout
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        emit(line)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

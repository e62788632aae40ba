"""Time the UNMODIFIED reference's training steps on this container's host cores next to the CPU oracle port (oracle/jg_oracle.py), same
weights, same batch, batch 1, fp32:
  palette  BASELINE configs[1] shape (256x256, efficient UNet, AdamW + EMA)                     -> profiles/r02_cpu_reference_vs_port.json
  cut      BASELINE configs[2] shape (cut_model, segformer_attn_conv G + [projected_d (vitsmall, proj_interp 256), basic] + mlp_sample F,
           MoNCE, 256x256): what bench.py's `cut` leg times.  timm is absent: `timm.create_model` returns oracle/vit_small_torch.py's
           restatement of vit_small_patch16_224, as in oracle/make_golden_projd_vit.py; everything else is the reference's own code
  cm       BASELINE configs[4] shape (cm_model consistency step, 256x256, efficient UNet, AdamW + EMA)
                                                                  cut + cm (round 6)            -> profiles/r06_cpu_reference_vs_port.json

TEST INFRASTRUCTURE ONLY (build container; the reference does not exist on the GPU box).
    PYTHONDONTWRITEBYTECODE=1 python oracle/measure_reference_cpu.py [palette] [cut] [cm]        (default: cut cm)
bench.py's `cpu_baseline` leg times the PORT on the GPU box's cores; this file is the once-measured ratio between the port and the
reference itself that the bench line quotes."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
from make_golden import build_opt, synth_batch  # noqa: E402


def timed(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


def measure_cut(cores, n=4):
    """reference CUTModel.optimize_parameters() (G / F group, then the D group: 4 optimizers, EMA) against OracleCUTTrainer.step on the same
    weights (the reference's own initialisation) and the same batch"""
    import random

    import timm
    from vit_small_torch import VitSmallPatch16

    timm.create_model = lambda name, img_size=224, pretrained=False, **kw: VitSmallPatch16(img_size)
    import make_golden_cutstep as mc
    from models import create_model

    S, B = 256, 1
    c = dict(netG="segformer_attn_conv", ngf=64, n_blocks=9, ndf=64, S=S, B=B, nce_layers="0,1,2,3", num_patches=256, nce_loss="monce", pool=50)
    opt = mc.build_opt(c)
    opt.D_netDs = ["projected_d", "basic"]
    opt.D_proj_network_type, opt.D_proj_interp = "vitsmall", 256
    torch.manual_seed(0)
    model = create_model(opt, 0)
    model.setup(opt)
    data = mc.batch(B, S, 500)
    model.data_dependent_initialize(data)
    sd = lambda net: {k: v.detach().clone() for k, v in net.state_dict().items()}
    sdG, sdF, sdD, sdPD = sd(model.netG_A), sd(model.netF), sd(model.netD_B_basic), sd(model.netD_B_projected_d)

    def run_ref():
        model.set_input(data)
        model.optimize_parameters()

    t_ref = timed(run_ref, n)
    tr = O.OracleCUTTrainer(sdG, sdF, sdD, 9, [0, 1, 2, 3], num_patches=256, T=opt.alg_cut_nce_T, monce=True, pool_size=50, pool_rng=random.Random(0),
                            ema_beta=0.999, gen="segformer", sdPD=sdPD, proj_interp=256)
    gen = torch.Generator().manual_seed(3)
    hw = [(S // k) ** 2 for k in (4, 8, 16, 32)]

    def run_port():
        ids = [[torch.randperm(m, generator=gen)[:min(256, m)] for m in hw] for _ in range(2)]
        uni = [torch.rand(2 * B, generator=gen) for _ in range(14)] + [torch.rand(2 * B, 256, generator=gen) for _ in range(2)] \
            + [torch.rand(B, generator=gen) for _ in range(56)]
        tr.step(data["A"], data["B"], ids[0], ids[1], uniforms=uni)

    t_port = timed(run_port, n)
    return dict(config="cut_model, segformer_attn_conv G + D_netDs [projected_d (vitsmall architecture via oracle/vit_small_torch.py, proj_interp 256), basic] + "
                       "mlp_sample F, MoNCE, 256x256, batch 1, Adam x4 + EMA, fp32 (BASELINE configs[2] shape; bench.py `cut` leg)",
                cores=cores, steps_timed=n, reference_s_per_step=round(t_ref, 4), port_s_per_step=round(t_port, 4),
                reference_img_per_s=round(B / t_ref, 4), port_img_per_s=round(B / t_port, 4), port_over_reference=round(t_ref / t_port, 3))


def measure_cm(cores, n=4):
    """reference CMModel.optimize_parameters() (student forward, no-grad teacher forward, pseudo-Huber loss, backward, AdamW, EMA) against
    OracleCMTrainer on the same weights and batch"""
    import make_golden_cm as mcm
    from models import create_model

    c = dict(ngf=64, mults=[1, 2, 4, 8], res_blocks=[2, 2, 2, 2], attn_res=[16], efficient=True, S=256, B=1)
    opt = mcm.build_opt(c)
    torch.manual_seed(0)
    model = create_model(opt, 0)
    model.setup(opt)
    model.use_temporal = False
    sd = {k: v.detach().clone() for k, v in model.netG_A.state_dict().items()}
    data = synth_batch(c["B"], c["S"], seed=99)

    def run_ref():
        model.set_input(data)
        model.optimize_parameters()

    t_ref = timed(run_ref, n)
    tr = O.OracleCMTrainer(sd, O.UNetCfg(in_channel=3, efficient=True, cond_embed_dim=getattr(opt, "alg_diffusion_cond_embed_dim", 256)), model.total_t)
    gen = torch.Generator().manual_seed(3)

    def run_port():
        noise, ts = O.cm_draw_step_randomness(gen, data["B"], tr.sigmas())
        tr.optimize_parameters(data["B"], data["B_label_mask"], noise, ts)

    t_port = timed(run_port, n)
    return dict(config="cm_model consistency step, efficient UNet ngf 64 mults [1,2,4,8], 256x256, batch 1, AdamW + EMA, fp32 (BASELINE configs[4] shape; bench.py `cm` leg)",
                cores=cores, steps_timed=n, reference_s_per_step=round(t_ref, 4), port_s_per_step=round(t_port, 4),
                reference_img_per_s=round(c["B"] / t_ref, 4), port_img_per_s=round(c["B"] / t_port, 4), port_over_reference=round(t_ref / t_port, 3))


def main():
    which = [a for a in sys.argv[1:] if a in ("palette", "cut", "cm")] or ["cut", "cm"]
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    os.chdir("/tmp")
    if "cut" in which or "cm" in which:
        out = {"torch": torch.__version__}
        path = os.path.join(ROOT, "profiles", "r06_cpu_reference_vs_port.json")
        if os.path.exists(path):
            out.update(json.load(open(path)))
        if "cut" in which:
            out["cut"] = measure_cut(cores)
            print(out["cut"], flush=True)
        if "cm" in which:
            out["cm"] = measure_cm(cores)
            print(out["cm"], flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    if "palette" in which:
        main_palette(cores)


def main_palette(cores):
    c = dict(ngf=64, mults=[1, 2, 4, 8], res_blocks=[2, 2, 2, 2], attn_res=[16], efficient=True, S=256, B=1)
    from models import create_model

    opt = build_opt(c)
    torch.manual_seed(0)
    model = create_model(opt, 0)
    model.setup(opt)
    model.use_temporal = False
    sd = {k: v.detach().clone() for k, v in model.netG_A.state_dict().items()}
    data = synth_batch(c["B"], c["S"], seed=99)
    n = 6

    def run_ref():
        model.set_input(data)
        model.optimize_parameters()

    run_ref()
    t0 = time.perf_counter()
    for _ in range(n):
        run_ref()
    t_ref = (time.perf_counter() - t0) / n
    tr = O.OraclePaletteTrainer(sd, O.UNetCfg(efficient=True))
    gen = torch.Generator().manual_seed(3)

    def run_port():
        t, u, noise = O.draw_step_randomness(gen, data["B"], 2000)
        tr.optimize_parameters(data["B"], data["A"], data["B_label_mask"], noise, t, u)

    run_port()
    t0 = time.perf_counter()
    for _ in range(n):
        run_port()
    t_port = (time.perf_counter() - t0) / n
    out = dict(config="palette_model DDPM, efficient UNet ngf 64 mults [1,2,4,8], 256x256, batch 1, AdamW + EMA, fp32", cores=cores,
               torch=torch.__version__, steps_timed=n, reference_s_per_step=round(t_ref, 4), port_s_per_step=round(t_port, 4),
               reference_img_per_s=round(c["B"] / t_ref, 4), port_img_per_s=round(c["B"] / t_port, 4),
               port_over_reference=round(t_ref / t_port, 3))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r02_cpu_reference_vs_port.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out)


if __name__ == "__main__":
    main()

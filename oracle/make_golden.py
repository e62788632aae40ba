"""Generate tests/golden/*.pt by running the UNMODIFIED reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container (the reference does not exist on the
GPU box):   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

What is pinned (all fp32, CPU, torch as installed here):
  * schedule buffers of the DDPM UNet (7 per phase)                     -> schedule.pt
  * UNet.forward + backward on tiny configs (efficient on/off)          -> unet_<cfg>.pt
  * DiffusionGenerator.forward given (t, u, noise)                      -> diffgen_<cfg>.pt
  * 3 x PaletteModel.optimize_parameters() (AdamW + EMA) from the
    options/JSON path (examples/example_ddpm_noglasses2glasses.json +
    overrides), with per-parameter projection checksums                 -> palette_step_<cfg>.pt
Weights are NOT stored: they are re-derived from (key, shape, seed) by
oracle/jg_oracle.synth_state_dict, so the fixtures stay a few hundred KB.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate

TINY = {
    # name: (G overrides, crop, batch)
    "tiny_eff": dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[16], efficient=True, S=16, B=2),
    "tiny_noeff": dict(ngf=32, mults=[1, 2, 2], res_blocks=[2, 1, 1], attn_res=[16], efficient=False, S=16, B=2),
    "tiny_attn": dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[2], efficient=True, S=16, B=1),
}


def build_opt(c):
    from options.train_options import TrainOptions
    import train as ref_train

    cfg = json.load(open(os.path.join(ref_shim.REFERENCE_ROOT, "examples/example_ddpm_noglasses2glasses.json")))
    cfg["data"]["crop_size"] = c["S"]
    cfg["data"]["load_size"] = c["S"]
    cfg["train"]["batch_size"] = c["B"]
    cfg["train"]["iter_size"] = 1
    cfg["gpu_ids"] = "-1"
    cfg["G"]["ngf"] = c["ngf"]
    cfg["G"]["unet_mha_channel_mults"] = c["mults"]
    cfg["G"]["unet_mha_res_blocks"] = c["res_blocks"]
    cfg["G"]["unet_mha_attn_res"] = c["attn_res"]
    cfg["G"]["unet_mha_vit_efficient"] = c["efficient"]
    cfg["output"]["display"]["type"] = ["none"]
    cfg["checkpoints_dir"] = "/tmp/jg_golden_ckpt/"
    cfg["dataroot"] = "/tmp/nodata"
    opt = TrainOptions().parse_json(cfg, save_config=False)
    opt.use_cuda = False
    opt.optim = ref_train.optim
    opt.jg_dir = ref_shim.REFERENCE_ROOT
    opt.total_iters = 0
    opt.num_test_images = 0
    return opt


def synth_batch(B, S, seed):
    """SURVEY.md §8(d) synthetic batch: B ~ U(-1,1), rectangle mask, A = B(1-m) + N(0,1) m."""
    g = torch.Generator().manual_seed(seed)
    Bimg = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    m = torch.zeros(B, 1, S, S, dtype=torch.int64)
    for i in range(B):
        h0 = int(torch.randint(0, S // 2, (1,), generator=g))
        w0 = int(torch.randint(0, S // 2, (1,), generator=g))
        hh = int(torch.randint(S // 4, S // 2 + 1, (1,), generator=g))
        ww = int(torch.randint(S // 4, S // 2 + 1, (1,), generator=g))
        m[i, :, h0:h0 + hh, w0:w0 + ww] = 1
    A = Bimg * (1 - m) + torch.randn(B, 3, S, S, generator=g) * m
    return {"A": A, "B": Bimg, "B_label_mask": m, "A_img_paths": ["synthetic"] * B}


def checks(named_tensors):
    """name -> (l2 norm, projection on a fixed pseudo-random vector)."""
    out = {}
    for k, v in named_tensors.items():
        v = v.detach().float()
        out[k] = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    from models import create_model

    # ---- schedule buffers -------------------------------------------------------------
    opt = build_opt(TINY["tiny_eff"])
    model = create_model(opt, 0)
    sd = model.netG_A.state_dict()
    sched = {k.split(".")[-1]: v.clone() for k, v in sd.items() if O._is_buffer(k)}
    assert len(sched) == 14, sorted(sched)
    torch.save(sched, os.path.join(OUT, "schedule.pt"))

    for name, c in TINY.items():
        opt = build_opt(c)
        torch.manual_seed(0)
        model = create_model(opt, 0)
        model.setup(opt)
        model.use_temporal = False
        netG = model.netG_A
        ref_sd = netG.state_dict()
        syn = O.synth_state_dict(ref_sd, seed=0)
        netG.load_state_dict(syn)
        unet = netG.denoise_fn.model
        B, S = c["B"], c["S"]

        # ---- UNet forward/backward ----------------------------------------------------
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, 6, S, S, generator=g)
        emb = torch.randn(B, 32, generator=g)
        R = torch.randn(B, 3, S, S, generator=g)
        x.requires_grad_(True)
        emb.requires_grad_(True)
        netG.zero_grad()
        out = unet(x, emb)
        (out * R).sum().backward()
        grads = {k: p.grad for k, p in unet.named_parameters()}
        torch.save(
            dict(cfg=c, x=x.detach(), emb=emb.detach(), R=R, out=out.detach(), dx=x.grad.clone(),
                 demb=emb.grad.clone(), grad_checks=checks(grads)),
            os.path.join(OUT, f"unet_{name}.pt"),
        )
        netG.zero_grad()

        # ---- DiffusionGenerator.forward with pinned randomness -----------------------
        data = synth_batch(B, S, seed=1234)
        y_0, y_cond, mask = data["B"], data["A"], data["B_label_mask"]
        gen = torch.Generator().manual_seed(77)
        t, u, noise = O.draw_step_randomness(gen, y_0, 2000)
        torch.manual_seed(77)  # the reference draws from the default generator, same order
        with torch.no_grad():
            n_ref, noise_hat, w = netG(y_0=y_0, y_cond=y_cond, mask=mask, noise=None, cls=None, ref=None)
        assert torch.equal(n_ref, noise), "default-generator draw order differs from draw_step_randomness"
        torch.save(
            dict(cfg=c, A=data["A"], B=data["B"], mask=mask, t=t, u=u, noise=noise,
                 noise_hat=noise_hat, min_snr_w=w),
            os.path.join(OUT, f"diffgen_{name}.pt"),
        )

        # ---- 3 full optimize_parameters() steps ---------------------------------------
        steps = []
        for it in range(3):
            data = synth_batch(B, S, seed=1234 + it)
            gen = torch.Generator().manual_seed(1000 + it)
            t, u, noise = O.draw_step_randomness(gen, data["B"], 2000)
            model.set_input(data)
            torch.manual_seed(1000 + it)
            model.optimize_parameters()
            loss = model.get_current_losses()["G_tot"].detach().clone()
            rec = dict(A=data["A"], B=data["B"], mask=data["B_label_mask"], t=t, u=u, noise=noise, loss=loss)
            if it in (0, 2):
                rec["param_checks"] = checks(dict(model.netG_A.named_parameters()))
                rec["ema_checks"] = checks(dict(model.netG_A_ema.named_parameters()))
            steps.append(rec)
            print(name, "step", it, "loss", float(loss))
        hp = dict(lr=opt.train_G_lr, beta1=opt.train_beta1, beta2=opt.train_beta2, eps=opt.train_optim_eps,
                  weight_decay=opt.train_optim_weight_decay, ema_beta=opt.train_G_ema_beta,
                  lambda_G=opt.alg_diffusion_lambda_G, optim=opt.train_optim)
        # a handful of raw parameter values after step 3 (first 8 entries of a few tensors)
        names = list(dict(model.netG_A.named_parameters()).keys())
        sample = {k: dict(model.netG_A.named_parameters())[k].detach().flatten()[:8].clone()
                  for k in names[:: max(1, len(names) // 12)]}
        torch.save(dict(cfg=c, hp=hp, steps=steps, param_sample=sample, keys=list(ref_sd.keys()),
                        shapes={k: tuple(v.shape) for k, v in ref_sd.items()}),
                   os.path.join(OUT, f"palette_step_{name}.pt"))

    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes:", tot)


if __name__ == "__main__":
    main()

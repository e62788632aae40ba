"""Generate tests/golden/cm_*.pt by running the UNMODIFIED reference (/root/reference) cm_model on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container:   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_cm.py

What is pinned (fp32, CPU):
  * CMGenerator.forward (ft_mode "cm") given the two random draws (noise, timesteps)   -> cm_gen_<cfg>.pt
  * 3 x CMModel.optimize_parameters() (AdamW + EMA) from examples/example_cm_noglasses2glasses.json
    + overrides, with per-parameter projection checksums                               -> cm_step_<cfg>.pt
Weights are re-derived from (key, shape, seed) by jg_oracle.synth_state_dict.
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
from make_golden import checks, synth_batch  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate

TINY = {
    "tiny_eff": dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[16], efficient=True, S=16, B=2),
    "tiny_attn": dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[2], efficient=False, S=16, B=2),
}


def build_opt(c):
    from options.train_options import TrainOptions
    import train as ref_train

    cfg = json.load(open(os.path.join(ref_shim.REFERENCE_ROOT, "examples/example_cm_noglasses2glasses.json")))
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


SAMPLING_SIGMAS = (80.0, 24.4, 5.84, 0.9, 0.661)      # cm_model.py:521


def make_sampling():
    """CMGenerator.restoration (cm_generator.py:504-554) of the unmodified reference, multistep consistency sampling over the sigmas
    CMModel.inference uses, with the N(0,1) draws recorded (the reference draws one randn_like per sigma from the default
    generator)                                                                          -> cm_sampling_<cfg>.pt"""
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    from models import create_model

    for name, c in TINY.items():
        opt = build_opt(c)
        torch.manual_seed(0)
        model = create_model(opt, 0)
        model.setup(opt)
        netG = model.netG_A
        netG.load_state_dict(O.synth_state_dict(netG.state_dict(), seed=0))
        netG.eval()
        data = synth_batch(c["B"], c["S"], seed=977)
        y_t, mask = data["A"], data["B_label_mask"]
        g = torch.Generator().manual_seed(31)
        noises = [torch.randn(y_t.shape, generator=g) for _ in SAMPLING_SIGMAS]
        it = iter(noises)
        orig = torch.randn_like
        torch.randn_like = lambda t, **kw: next(it).to(t.dtype)       # the reference's draws, in its order
        try:
            with torch.no_grad():
                out = netG.restoration(y_t, None, SAMPLING_SIGMAS, mask)
                out_noclip = None
        finally:
            torch.randn_like = orig
        assert next(it, None) is None
        torch.save(dict(cfg=c, y_t=y_t, mask=mask, sigmas=SAMPLING_SIGMAS, noises=noises, output=out),
                   os.path.join(OUT, f"cm_sampling_{name}.pt"))
        print(name, "restoration", tuple(out.shape), float(out.abs().mean()))


def main():
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    from models import create_model

    for name, c in TINY.items():
        opt = build_opt(c)
        assert opt.model_type == "cm", opt.model_type
        torch.manual_seed(0)
        model = create_model(opt, 0)
        model.setup(opt)
        model.use_temporal = False
        netG = model.netG_A
        ref_sd = netG.state_dict()
        netG.load_state_dict(O.synth_state_dict(ref_sd, seed=0))
        B, S = c["B"], c["S"]
        total_t = model.total_t

        # ---- CMGenerator.forward with pinned randomness -------------------------------------------------
        data = synth_batch(B, S, seed=4321)
        y_0, mask = data["B"], data["B_label_mask"]
        netG.current_t = 0
        sig = O.cm_karras_schedule(O.cm_improved_timesteps_schedule(0, total_t))
        noise, timesteps = O.cm_draw_step_randomness(torch.Generator().manual_seed(55), y_0, sig)
        torch.manual_seed(55)   # the reference draws from the default generator, same order
        with torch.no_grad():
            out = netG(y_0, total_t, mask, None)
        next_x, current_x, num_timesteps, sigmas, loss_weights, next_noisy_x, current_noisy_x = out
        assert torch.equal(sigmas, sig), "karras schedule restatement differs"
        chk = y_0 + sigmas[timesteps + 1].view(-1, 1, 1, 1) * noise
        m = torch.clamp(mask, min=0.0, max=1.0)
        assert torch.allclose(next_noisy_x, chk * m + (1 - m) * y_0), "draw order differs from cm_draw_step_randomness"
        torch.save(dict(cfg=c, total_t=total_t, B=y_0, mask=mask, noise=noise, timesteps=timesteps, next_x=next_x,
                        current_x=current_x, num_timesteps=num_timesteps, sigmas=sigmas, loss_weights=loss_weights,
                        next_noisy_x=next_noisy_x, current_noisy_x=current_noisy_x),
                   os.path.join(OUT, f"cm_gen_{name}.pt"))

        # ---- 3 full optimize_parameters() steps -----------------------------------------------------------
        netG.current_t = 0
        steps = []
        cur_t = 0
        for it in range(3):
            data = synth_batch(B, S, seed=4321 + it)
            sig = O.cm_karras_schedule(O.cm_improved_timesteps_schedule(cur_t, total_t))
            noise, timesteps = O.cm_draw_step_randomness(torch.Generator().manual_seed(2000 + it), data["B"], sig)
            model.set_input(data)
            torch.manual_seed(2000 + it)
            model.optimize_parameters()
            cur_t += B
            loss = model.get_current_losses()["G_tot"].detach().clone()
            rec = dict(A=data["A"], B=data["B"], mask=data["B_label_mask"], noise=noise, timesteps=timesteps, loss=loss)
            if it in (0, 2):
                rec["param_checks"] = checks(dict(model.netG_A.named_parameters()))
                rec["ema_checks"] = checks(dict(model.netG_A_ema.named_parameters()))
            steps.append(rec)
            print(name, "step", it, "loss", float(loss))
        hp = dict(lr=opt.train_G_lr, beta1=opt.train_beta1, beta2=opt.train_beta2, eps=opt.train_optim_eps,
                  weight_decay=opt.train_optim_weight_decay, ema_beta=opt.train_G_ema_beta,
                  lambda_G=opt.alg_diffusion_lambda_G, optim=opt.train_optim, ema=bool(opt.train_G_ema))
        torch.save(dict(cfg=c, hp=hp, total_t=total_t, steps=steps, keys=list(ref_sd.keys()),
                        shapes={k: tuple(v.shape) for k, v in ref_sd.items()}),
                   os.path.join(OUT, f"cm_step_{name}.pt"))
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.startswith("cm_"))
    print("cm golden bytes:", tot)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sampling":     # only cm_sampling_*.pt
        make_sampling()
    else:                                                   # every cm_*.pt fixture
        main()
        make_sampling()

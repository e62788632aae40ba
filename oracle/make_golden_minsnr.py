"""Generate tests/golden/palette_step_minsnr_tiny.pt by running the UNMODIFIED reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container:   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_minsnr.py

`alg_palette_minsnr = True` -- what the reference's own run tests select (tests/test_run_diffusion.py:31) -- : the min-SNR-gamma weight
min(SNR, 5) / SNR of every sample multiplies both operands of the loss (models/palette_model.py:586-620,
models/modules/diffusion_generator.py:493-527).  3 x PaletteModel.optimize_parameters() (inpainting) from the options / JSON path.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
import make_golden as MG  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate
CFG = dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[16], efficient=True, S=16, B=2)


def main():
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    import json

    from models import create_model
    from options.train_options import TrainOptions
    import train as ref_train

    c = CFG
    cfg = json.load(open(os.path.join(ref_shim.REFERENCE_ROOT, "examples/example_ddpm_noglasses2glasses.json")))
    cfg["data"]["crop_size"] = cfg["data"]["load_size"] = c["S"]
    cfg["train"]["batch_size"], cfg["train"]["iter_size"] = c["B"], 1
    cfg["gpu_ids"] = "-1"
    cfg["G"].update(ngf=c["ngf"], unet_mha_channel_mults=c["mults"], unet_mha_res_blocks=c["res_blocks"], unet_mha_attn_res=c["attn_res"],
                    unet_mha_vit_efficient=c["efficient"])
    cfg["alg"]["palette"]["minsnr"] = True
    cfg["output"]["display"]["type"] = ["none"]
    cfg["checkpoints_dir"], cfg["dataroot"] = "/tmp/jg_golden_ckpt/", "/tmp/nodata"
    opt = TrainOptions().parse_json(cfg, save_config=False)
    opt.use_cuda, opt.optim, opt.jg_dir, opt.total_iters, opt.num_test_images = False, ref_train.optim, ref_shim.REFERENCE_ROOT, 0, 0
    torch.manual_seed(0)
    model = create_model(opt, 0)
    model.setup(opt)
    model.use_temporal = False
    ref_sd = model.netG_A.state_dict()
    model.netG_A.load_state_dict(O.synth_state_dict(ref_sd, seed=0))
    steps = []
    for it in range(3):
        data = MG.synth_batch(c["B"], c["S"], seed=5321 + it)
        gen = torch.Generator().manual_seed(2500 + it)
        t, u, noise = O.draw_step_randomness(gen, data["B"], 2000)
        model.set_input(data)
        torch.manual_seed(2500 + it)
        model.optimize_parameters()
        loss = model.get_current_losses()["G_tot"].detach().clone()
        rec = dict(A=data["A"], B=data["B"], mask=data["B_label_mask"], t=t, u=u, noise=noise, loss=loss)
        if it in (0, 2):
            rec["param_checks"] = MG.checks(dict(model.netG_A.named_parameters()))
            rec["ema_checks"] = MG.checks(dict(model.netG_A_ema.named_parameters()))
        steps.append(rec)
        print("minsnr step", it, "loss", float(loss))
    hp = dict(lr=opt.train_G_lr, beta1=opt.train_beta1, beta2=opt.train_beta2, eps=opt.train_optim_eps, weight_decay=opt.train_optim_weight_decay,
              ema_beta=opt.train_G_ema_beta, lambda_G=opt.alg_diffusion_lambda_G, optim=opt.train_optim)
    assert opt.alg_palette_minsnr is True
    torch.save(dict(cfg=c, hp=hp, minsnr=True, steps=steps, keys=list(ref_sd.keys()), shapes={k: tuple(v.shape) for k, v in ref_sd.items()}),
               os.path.join(OUT, "palette_step_minsnr_tiny.pt"))


if __name__ == "__main__":
    main()

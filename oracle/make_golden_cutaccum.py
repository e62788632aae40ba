"""Generate tests/golden/cutstep_accum.pt by running the UNMODIFIED reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container:   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_cutaccum.py

Gradient accumulation of the CUT step, `train_iter_size = 2` (the shipped examples/example_gan_horse2zebra.json trains with 8,
example_gan_noglasses2glasses.json with 16): four calls of CUTModel.optimize_parameters() = two optimizer steps of every network.
models/base_model.py:1250-1282,1302-1377: per call and per group (G/F, then D) the group's loss / iter_size is back-propagated into the
accumulated gradients of the group's networks only (the other networks have requires_grad False); the group's optimizers step and clear
them when niter % iter_size == 0; the EMA of G_A is updated on EVERY call; get_current_losses() reports the `<name>_avg` values
published at the boundary.  Patch ids, pool draws recorded like oracle/make_golden_cutstep.py.
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
import make_golden_cutstep as CS  # noqa: E402
from make_golden import checks  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate
CFG = dict(ngf=16, n_blocks=2, ndf=16, S=32, B=2, nce_layers="0,4,8,10,11", num_patches=64, nce_loss="monce", pool=2, iters=4)
ITER_SIZE = 2
RAW = ("G_tot", "G_NCE", "G_NCE_Y", "G_GAN_D_B_basic", "D_tot", "D_GAN_D_B_basic")


def main():
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    import util.image_pool as ref_pool
    from models import create_model

    c = CFG
    opt = CS.build_opt(c)
    opt.train_iter_size = ITER_SIZE          # read by the model constructor (iter_calculator_init) and by every optimize_parameters()
    torch.manual_seed(0)
    model = create_model(opt, 0)
    model.setup(opt)
    rr = CS.RecordingRandom(99)
    ref_pool.random = rr
    perms = []
    real_randperm = torch.randperm

    def rec_randperm(*a, **k):
        p = real_randperm(*a, **k)
        perms.append(p.clone())
        return p

    torch.randperm = rec_randperm
    try:
        model.data_dependent_initialize(CS.batch(c["B"], c["S"], 500))
        sdG = O.synth_state_dict(model.netG_A.state_dict(), seed=0)
        sdD = O.synth_state_dict(model.netD_B_basic.state_dict(), seed=1)
        sdF = O.synth_state_dict(model.netF.state_dict(), seed=3)
        model.netG_A.load_state_dict(sdG)
        model.netD_B_basic.load_state_dict(sdD)
        model.netF.load_state_dict(sdF)
        steps = []
        for it in range(c["iters"]):
            data = CS.batch(c["B"], c["S"], 500 + it)
            model.set_input(data)
            perms.clear()
            n_log = len(rr.log)
            torch.manual_seed(100 + it)
            model.optimize_parameters()
            rec = dict(A=data["A"], B=data["B"], perms=[p[: c["num_patches"]].clone() for p in perms], pool_draws=list(rr.log[n_log:]),
                       raw={k: float(getattr(model, "loss_" + k)) for k in RAW}, fake_B=model.fake_B.detach().clone(),
                       G_checks=checks(dict(model.netG_A.named_parameters())), F_checks=checks(dict(model.netF.named_parameters())),
                       D_checks=checks(dict(model.netD_B_basic.named_parameters())),
                       ema_checks=checks(dict(model.netG_A_ema.named_parameters())))
            if (it + 1) % ITER_SIZE == 0:
                rec["losses_reported"] = {k: float(v) for k, v in model.get_current_losses().items()}
            steps.append(rec)
            print("cut accum call", it, {k: round(v, 5) for k, v in rec["raw"].items()}, rec.get("losses_reported"))
    finally:
        torch.randperm = real_randperm
        ref_pool.random = random
    hp = dict(lr_G=opt.train_G_lr, lr_D=opt.train_D_lr, beta1=opt.train_beta1, beta2=opt.train_beta2, eps=opt.train_optim_eps,
              ema_beta=opt.train_G_ema_beta, T=opt.alg_cut_nce_T, lambda_NCE=opt.alg_cut_lambda_NCE, lambda_GAN=opt.alg_gan_lambda)
    torch.save(dict(cfg=c, hp=hp, iter_size=ITER_SIZE, steps=steps, keysG=list(sdG.keys()), shapesG={k: tuple(v.shape) for k, v in sdG.items()},
                    keysD=list(sdD.keys()), shapesD={k: tuple(v.shape) for k, v in sdD.items()}, keysF=list(sdF.keys()),
                    shapesF={k: tuple(v.shape) for k, v in sdF.items()}, loss_names=list(model.loss_names)),
               os.path.join(OUT, "cutstep_accum.pt"))


if __name__ == "__main__":
    main()

"""Generate tests/golden/cutloss_*.pt and cutstep_*.pt from the UNMODIFIED reference on CPU (TEST INFRASTRUCTURE ONLY):
  * cutloss_<cfg>.pt : PatchSampleF (+MLP), PatchNCELoss and MoNCELoss (50 Sinkhorn iterations, differentiated through), GANLoss
                       'lsgan' -- outputs and input gradients on seeded inputs;
  * cutstep_<cfg>.pt : N x CUTModel.optimize_parameters() (models/cut_model.py through the options/JSON path: resnet generator,
                       'basic' PatchGAN, mlp_sample netF, MoNCE / PatchNCE, nce_idt, lsgan, Adam x3, EMA, image pool) with the
                       torch.randperm patch ids and the python-`random` pool draws RECORDED, losses and parameter checksums.
   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_cutstep.py"""
import json
import os
import random
import sys
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402

REAL_RAND = torch.rand        # step_fixtures() swaps torch.rand for the SegFormer uniform recorder while a model runs
from make_golden import checks  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate
STEP_CFGS = {
    "monce": dict(ngf=16, n_blocks=2, ndf=16, S=32, B=2, nce_layers="0,4,8,10,11", num_patches=64, nce_loss="monce", pool=2, iters=4),
    # BASELINE.json configs[0]: cut_model, resnet_9blocks G + basic D, 128x128, batch 1 (example_gan_horse2zebra.json shape)
    "config0": dict(ngf=64, n_blocks=9, ndf=64, S=128, B=1, nce_layers="0,4,8,12,16", num_patches=256, nce_loss="monce", pool=50, iters=2),
    # BASELINE.json configs[2] shape with the buildable discriminator: SegFormer-attn G (MiT-b0 + ResnetDecoder tail) + basic D + MoNCE
    # batch_rand="recorder": the committed segformer / mobile_attn fixtures were generated AFTER the uniform recorder was added, with
    # batch() drawing A / B through the recorder's torch.rand (its generator, seed 31) instead of Generator(seed); the three older
    # fixtures drew them with the real torch.rand.  The key pins which one each fixture used, so that every committed file regenerates
    # bit-exact from this script (tests/test_oracle_golden.py::test_fixtures_regenerate); the inputs are stored in the fixture either way.
    "segformer": dict(netG="segformer_attn_conv", ngf=64, n_blocks=9, ndf=16, S=64, B=2, nce_layers="0,1,2,3", num_patches=64, nce_loss="monce",
                      pool=2, iters=3, batch_rand="recorder"),
    # attention ResNet generator with depth-wise separable blocks (G_netG = mobile_resnet_attn); nce ids >= n_blocks tap nothing
    "mobile_attn": dict(netG="mobile_resnet_attn", ngf=16, n_blocks=3, ndf=16, S=64, B=2, nce_layers="0,1,2,8", num_patches=64, nce_loss="monce",
                        pool=2, iters=3, batch_rand="recorder"),
    "patchnce": dict(ngf=16, n_blocks=3, ndf=16, S=32, B=1, nce_layers="0,4,8,12", num_patches=32, nce_loss="patchnce", pool=1, iters=3),
}


class RecordingRandom:
    """stands in for the `random` module inside util.image_pool; records every draw."""

    def __init__(self, seed):
        self.r, self.log = random.Random(seed), []

    def uniform(self, a, b):
        v = self.r.uniform(a, b)
        self.log.append(("uniform", v))
        return v

    def randint(self, a, b):
        v = self.r.randint(a, b)
        self.log.append(("randint", v))
        return v


def loss_fixtures():
    torch.manual_seed(0)
    from models.modules.NCE.monce import MoNCELoss
    from models.modules.NCE.patchnce import PatchNCELoss
    from models.modules.cut_networks import PatchSampleF
    from models.modules.loss import GANLoss

    for name, (B, P, chans, sizes) in {"a": (2, 64, [3, 32, 64], [12, 8, 8]), "b": (1, 256, [128], [16])}.items():
        g = torch.Generator().manual_seed(21)
        feats = [torch.randn(B, c, s, s, generator=g) for c, s in zip(chans, sizes)]
        netF = PatchSampleF(use_mlp=True, init_type="normal", init_gain=0.02, nc=256)
        netF.set_device(torch.device("cpu"))
        netF.data_dependent_initialize(feats)
        sdF = O.synth_state_dict(netF.state_dict(), seed=3)
        netF.load_state_dict(sdF)
        fk = [f.clone().requires_grad_(True) for f in feats]
        fq = [(f + 0.3 * torch.randn(f.shape, generator=g)).requires_grad_(True) for f in feats]
        k_pool, ids = netF(fk, P, None)
        q_pool, _ = netF(fq, P, ids)
        rec = dict(B=B, P=P, feats_k=[f.detach() for f in fk], feats_q=[f.detach() for f in fq], ids=[i.clone() for i in ids],
                   k_pool=[k.detach() for k in k_pool], q_pool=[q.detach() for q in q_pool], keysF=list(sdF.keys()),
                   shapesF={k: tuple(v.shape) for k, v in sdF.items()})
        opt = SimpleNamespace(alg_cut_nce_includes_all_negatives_from_minibatch=False, alg_cut_nce_T=0.07, alg_cut_num_patches=P)
        for lname, cls in (("monce", MoNCELoss), ("patchnce", PatchNCELoss)):
            netF.zero_grad()
            for f in fk + fq:
                f.grad = None
            k_pool, _ = netF(fk, P, ids)
            q_pool, _ = netF(fq, P, ids)
            crit = cls(opt)
            per = [crit(feat_q=q, feat_k=k, current_batch=B) for q, k in zip(q_pool, k_pool)]
            # also the raw gradients with respect to the pooled features of the first layer
            gq, gk = torch.autograd.grad(per[0].mean(), [q_pool[0], k_pool[0]], retain_graph=True)
            total = sum(p.mean() for p in per) / len(per)
            total.backward()
            rec[lname] = dict(per=[p.detach() for p in per], total=total.detach(), dq0=gq, dk0=gk,
                              dfeats_q=[f.grad.clone() for f in fq], dfeats_k=[f.grad.clone() for f in fk],
                              gradF=checks({k: p.grad for k, p in netF.named_parameters()}))
        pred = torch.randn(B, 1, 6, 6, generator=g, requires_grad=True)
        gan = GANLoss("lsgan")
        l1 = gan(pred, True)
        (g1,) = torch.autograd.grad(l1, pred)
        l0 = gan(pred, False)
        (g0,) = torch.autograd.grad(l0, pred)
        rec["lsgan"] = dict(pred=pred.detach(), real=l1.detach(), dreal=g1, fake=l0.detach(), dfake=g0)
        torch.save(rec, os.path.join(OUT, f"cutloss_{name}.pt"))
        print("cutloss", name, [tuple(q.shape) for q in rec["q_pool"]], float(rec["monce"]["total"]), float(rec["patchnce"]["total"]))


def build_opt(c):
    from options.train_options import TrainOptions
    import train as ref_train

    cfg = json.load(open(os.path.join(ref_shim.REFERENCE_ROOT, "examples/example_gan_noglasses2glasses.json")))
    cfg["data"]["crop_size"] = cfg["data"]["load_size"] = c["S"]
    cfg["train"]["batch_size"], cfg["train"]["iter_size"] = c["B"], 1
    cfg["train"]["pool_size"] = c["pool"]
    cfg["train"]["semantic_mask"] = False
    cfg["train"]["mask"]["out_mask"] = False
    cfg["gpu_ids"] = "-1"
    cfg["G"].update(netG=c.get("netG", "resnet"), ngf=c["ngf"], nblocks=c["n_blocks"])
    cfg["D"].update(netDs=["basic"], ndf=c["ndf"])
    cfg["alg"]["cut"].update(nce_layers=c["nce_layers"], num_patches=c["num_patches"], nce_loss=c["nce_loss"])
    cfg["output"]["display"]["type"] = ["none"]
    cfg["checkpoints_dir"], cfg["dataroot"] = "/tmp/jg_golden_ckpt/", "/tmp/nodata"
    opt = TrainOptions().parse_json(cfg, save_config=False)
    opt.use_cuda = False
    opt.optim = ref_train.optim
    opt.jg_dir = ref_shim.REFERENCE_ROOT
    opt.total_iters = 0
    opt.num_test_images = 0
    return opt


def batch(B, S, seed, rand=REAL_RAND):
    g = torch.Generator().manual_seed(seed)
    return {"A": rand(B, 3, S, S, generator=g) * 2 - 1, "B": rand(B, 3, S, S, generator=g) * 2 - 1,
            "A_img_paths": ["synthetic"] * B, "B_img_paths": ["synthetic"] * B}


def step_fixtures():
    import util.image_pool as ref_pool
    from models import create_model

    real_randperm = torch.randperm
    for name, c in STEP_CFGS.items():
        if ONLY and name not in ONLY:
            continue
        opt = build_opt(c)
        torch.manual_seed(0)
        model = create_model(opt, 0)
        model.setup(opt)
        rr = RecordingRandom(99)
        ref_pool.random = rr
        perms = []

        def rec_randperm(*a, **k):
            p = real_randperm(*a, **k)
            perms.append(p.clone())
            return p

        torch.randperm = rec_randperm
        from make_golden_segformer import Recorder        # DropPath / Dropout2d uniforms of the SegFormer generator (none for resnet)
        import torch.nn.functional as F
        urec = Recorder(31)
        torch.rand, F.dropout2d = urec.rand, urec.dropout2d
        try:
            brand = urec.rand if c.get("batch_rand") == "recorder" else REAL_RAND
            data0 = batch(c["B"], c["S"], 500, brand)
            model.data_dependent_initialize(data0)
            sdG = O.synth_state_dict(model.netG_A.state_dict(), seed=0)
            sdD = O.synth_state_dict(model.netD_B_basic.state_dict(), seed=1)
            sdF = O.synth_state_dict(model.netF.state_dict(), seed=3)
            model.netG_A.load_state_dict(sdG)
            model.netD_B_basic.load_state_dict(sdD)
            model.netF.load_state_dict(sdF)
            steps = []
            for it in range(c["iters"]):
                data = batch(c["B"], c["S"], 500 + it, brand)
                model.set_input(data)
                perms.clear()
                n_log, n_u = len(rr.log), len(urec.log)
                torch.manual_seed(100 + it)
                model.optimize_parameters()
                losses = {k: float(v) for k, v in model.get_current_losses().items()}
                rec = dict(A=data["A"], B=data["B"], perms=[p[: c["num_patches"]].clone() for p in perms], pool_draws=list(rr.log[n_log:]), losses=losses,
                           fake_B=model.fake_B.detach().clone())
                if c.get("batch_rand") == "recorder":      # the fixtures written since the recorder exists carry the DropPath / Dropout2d uniforms
                    rec["uniforms"] = [u.clone() for u in urec.log[n_u:]]
                if it in (0, c["iters"] - 1):
                    rec["G_checks"] = checks(dict(model.netG_A.named_parameters()))
                    rec["F_checks"] = checks(dict(model.netF.named_parameters()))
                    rec["D_checks"] = checks(dict(model.netD_B_basic.named_parameters()))
                    rec["ema_checks"] = checks(dict(model.netG_A_ema.named_parameters()))
                steps.append(rec)
                print(name, it, {k: round(v, 5) for k, v in losses.items()}, "perms", [len(p) for p in perms], "draws", len(rec["pool_draws"]))
        finally:
            torch.rand, F.dropout2d = urec.real_rand, urec.real_d2
            torch.randperm = real_randperm
            ref_pool.random = random
        hp = dict(lr_G=opt.train_G_lr, lr_D=opt.train_D_lr, beta1=opt.train_beta1, beta2=opt.train_beta2, eps=opt.train_optim_eps,
                  ema_beta=opt.train_G_ema_beta, T=opt.alg_cut_nce_T, lambda_NCE=opt.alg_cut_lambda_NCE, lambda_GAN=opt.alg_gan_lambda)
        torch.save(dict(cfg={k: v for k, v in c.items() if k != "batch_rand"}, hp=hp, steps=steps, keysG=list(sdG.keys()), shapesG={k: tuple(v.shape) for k, v in sdG.items()},
                        keysD=list(sdD.keys()), shapesD={k: tuple(v.shape) for k, v in sdD.items()}, keysF=list(sdF.keys()),
                        shapesF={k: tuple(v.shape) for k, v in sdF.items()}, loss_names=list(model.loss_names)),
                   os.path.join(OUT, f"cutstep_{name}.pt"))


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]      # e.g. `config0` regenerates only that step fixture

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    if not ONLY:
        loss_fixtures()
    step_fixtures()
    print("bytes:", {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT)) if f.startswith("cut")})

"""Generate tests/golden/palette_{cls,mask}_tiny.pt: class- / mask-conditioned palette_model of the UNMODIFIED reference on CPU
(alg_diffusion_cond_embed = "class" | "mask", alg_diffusion_dropout_prob = 0.5 -> one extra "unconditioned" class):
  * DiffusionGenerator.forward with class labels given (t, u, noise)
  * 3 x PaletteModel.optimize_parameters() with the conditioning dropout (its uniform draw precedes t, u, noise on the default RNG)
  * DDPM restoration with class labels on a short test schedule.
TEST INFRASTRUCTURE ONLY.   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_cond.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
import make_golden as MG  # noqa: E402
from make_golden import checks  # noqa: E402

OUT = MG.OUT
CFG = dict(MG.TINY["tiny_eff"], nclasses=5, dropout_prob=0.5, cond_embed_dim=32)
T_TEST = 6


def build_opt(c):
    import json
    from options.train_options import TrainOptions
    import train as ref_train

    cfg = json.load(open(os.path.join(ref_shim.REFERENCE_ROOT, "examples/example_ddpm_noglasses2glasses.json")))
    cfg["data"]["crop_size"] = cfg["data"]["load_size"] = c["S"]
    cfg["train"]["batch_size"], cfg["train"]["iter_size"] = c["B"], 1
    cfg["gpu_ids"] = "-1"
    cfg["G"].update(ngf=c["ngf"], unet_mha_channel_mults=c["mults"], unet_mha_res_blocks=c["res_blocks"], unet_mha_attn_res=c["attn_res"],
                    unet_mha_vit_efficient=c["efficient"], diff_n_timestep_test=T_TEST)
    cfg["alg"]["diffusion"].update(cond_embed=c["cond"], cond_embed_dim=c["cond_embed_dim"], dropout_prob=c["dropout_prob"])
    cfg["f_s"]["semantic_nclasses"] = c["nclasses"]
    cfg.setdefault("cls", {})["semantic_nclasses"] = c["nclasses"]
    cfg["output"]["display"]["type"] = ["none"]
    cfg["checkpoints_dir"], cfg["dataroot"] = "/tmp/jg_golden_ckpt/", "/tmp/nodata"
    opt = TrainOptions().parse_json(cfg, save_config=False)
    opt.use_cuda = False
    opt.optim = ref_train.optim
    opt.jg_dir = ref_shim.REFERENCE_ROOT
    opt.total_iters = 0
    opt.num_test_images = 0
    return opt


def batch(B, S, seed, multi_class):
    d = MG.synth_batch(B, S, seed)
    if multi_class:      # semantic classes 1 .. 4 inside the rectangles (the blend clamps them to 1, the mask embedding sees the class)
        for i in range(B):
            d["B_label_mask"][i] *= 1 + (seed + i) % 4
    return d


def generate(tag, cond):
    from models import create_model

    c = dict(CFG, cond=cond)
    use_cls, use_mask = "class" in cond, "mask" in cond
    opt = build_opt(c)
    torch.manual_seed(0)
    model = create_model(opt, 0)
    model.setup(opt)
    model.use_temporal = False
    netG = model.netG_A
    ref_sd = netG.state_dict()
    tkey = "denoise_fn.netl_embedder_%s.embedding_table.weight" % ("class" if use_cls else "mask")
    assert ref_sd[tkey].shape[0] == c["nclasses"] + 1, ref_sd[tkey].shape          # after_parse added the unconditioned class
    syn = O.synth_state_dict(ref_sd, seed=0)
    syn[tkey][2] *= 8.0          # one row longer than max_norm: the in-place renormalisation is exercised
    netG.load_state_dict(syn)
    B, S = c["B"], c["S"]

    # ---- DiffusionGenerator.forward with the conditioning ------------------------------------------
    data = batch(B, S, 4321, use_mask)
    cls0 = torch.tensor([2, 4][:B], dtype=torch.int64) if use_cls else None
    gen = torch.Generator().manual_seed(55)
    t, u, noise = O.draw_step_randomness(gen, data["B"], 2000)
    torch.manual_seed(55)
    with torch.no_grad():
        n_ref, noise_hat, w = netG(y_0=data["B"], y_cond=data["A"], mask=data["B_label_mask"], noise=None, cls=cls0, ref=None)
    assert torch.equal(n_ref, noise)
    table_after = netG.state_dict()[tkey].clone()
    fwd = dict(A=data["A"], B=data["B"], mask=data["B_label_mask"], cls=cls0, t=t, u=u, noise=noise, noise_hat=noise_hat, min_snr_w=w,
               table_row2_norm_after=table_after[2].norm().clone())
    netG.load_state_dict(syn)

    # ---- 3 optimize_parameters() with conditioning dropout ----------------------------------------
    steps = []
    for it in range(3):
        data = batch(B, S, 2234 + it, use_mask)
        if use_cls:
            data["B_label_cls"] = torch.tensor([(1 + it) % c["nclasses"], (2 + 2 * it) % c["nclasses"]][:B], dtype=torch.int64)
        gen = torch.Generator().manual_seed(3000 + it)
        drop_u = torch.rand(B, generator=gen)            # compute_palette_loss draws it first (palette_model.py:565-571)
        t, u, noise = O.draw_step_randomness(gen, data["B"], 2000)
        model.set_input(data)
        torch.manual_seed(3000 + it)
        model.optimize_parameters()
        loss = model.get_current_losses()["G_tot"].detach().clone()
        rec = dict(A=data["A"], B=data["B"], mask=data["B_label_mask"], cls=data.get("B_label_cls"), drop_u=drop_u, t=t, u=u, noise=noise, loss=loss)
        if it in (0, 2):
            rec["param_checks"] = checks(dict(model.netG_A.named_parameters()))
            rec["ema_checks"] = checks(dict(model.netG_A_ema.named_parameters()))
        steps.append(rec)
        print(tag, "step", it, "loss", float(loss), "dropped", (drop_u < c["dropout_prob"]).tolist())
    hp = dict(lr=opt.train_G_lr, beta1=opt.train_beta1, beta2=opt.train_beta2, eps=opt.train_optim_eps, weight_decay=opt.train_optim_weight_decay,
              ema_beta=opt.train_G_ema_beta, lambda_G=opt.alg_diffusion_lambda_G, optim=opt.train_optim)

    # ---- DDPM restoration with the conditioning -----------------------------------------------------
    netG.load_state_dict(syn)
    netG.eval()
    data = batch(B, S, 888, use_mask)
    g = torch.Generator().manual_seed(41)
    y_t0 = torch.randn(B, 3, S, S, generator=g)
    gen = torch.Generator().manual_seed(42)
    noises = [torch.randn(B, 3, S, S, generator=gen) for _ in range(T_TEST - 1)]
    torch.manual_seed(42)
    with torch.no_grad():
        y_out, ret = netG.restoration(data["A"], y_t=y_t0.clone(), y_0=data["B"], mask=data["B_label_mask"], sample_num=2, cls=cls0)
    samp = dict(A=data["A"], B=data["B"], mask=data["B_label_mask"], cls=cls0, y_t0=y_t0, noises=noises, y_out=y_out, ret=ret, T=T_TEST)
    path = os.path.join(OUT, f"palette_{tag}_tiny.pt")
    torch.save(dict(cfg=c, hp=hp, fwd=fwd, steps=steps, sampling=samp, num_classes=model.num_classes, keys=list(ref_sd.keys()),
                    shapes={k: tuple(v.shape) for k, v in ref_sd.items()}, table_key=tkey, table_row_scale=(2, 8.0)), path)
    print(tag, "bytes", os.path.getsize(path), "num_classes", model.num_classes, "in_channel", ref_sd["denoise_fn.model.input_blocks.0.0.weight"].shape[1])


if __name__ == "__main__":
    os.chdir("/tmp")
    generate("cls", "class")
    generate("mask", "mask")

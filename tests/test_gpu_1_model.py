"""GPU parity tests, model level: the HIP UNet / DiffusionGenerator / PaletteModel step against
(1) the golden fixtures produced by the unmodified reference and (2) the CPU oracle on larger
seeded inputs.  All calls go through the C ABI (libjg355.so via ctypes)."""
import os

import pytest
import torch

import jg_oracle as O

pytestmark = pytest.mark.gpu

CFGS = ["tiny_eff", "tiny_noeff", "tiny_attn"]
# model-level tolerances (norm-wise relative): errors of ~60 16-bit layers compound
TOL_OUT = {torch.float16: 4e-3, torch.bfloat16: 3e-2}
TOL_GRAD = {torch.float16: 1.5e-2, torch.bfloat16: 8e-2}


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


EPS16 = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


def noise_floor(key, ref_norms, dtype):
    """Gradients that are analytically (near) zero: the bias of a conv whose output feeds a GroupNorm
    (in_layers.2.bias, skip_connection.bias, out_layers.3.bias ...) receives sum_p dy[p] where dy has
    (near) zero mean per group -- what is left is the rounding noise of the 16-bit dy, of the order
    eps16 * |dy|_1.  The reference's fp32 value is the same thing at fp32 eps (1e-6).  Such entries are
    compared with an absolute floor tied to the weight gradient of the same layer."""
    if not key.endswith(".bias"):
        return 0.0
    wk = key[:-4] + "weight"
    ref = ref_norms.get(wk)
    if ref is None:
        return 0.0
    r0 = float(ref[0]) if torch.is_tensor(ref) else float(ref)
    return 16.0 * EPS16[dtype] * r0


def load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def overrides_of(c, **extra):
    d = dict(G_ngf=c["ngf"], G_unet_mha_channel_mults=c["mults"], G_unet_mha_res_blocks=c["res_blocks"],
             G_unet_mha_attn_res=c["attn_res"], G_unet_mha_vit_efficient=c["efficient"], data_crop_size=c["S"],
             train_batch_size=c["B"])
    d.update(extra)
    return d


def cfg_of(c):
    extra = c.get("cond_embed_dim", 0) if "mask" in c.get("cond", "") else 0       # mask conditioning widens the UNet input
    return O.UNetCfg(in_channel=6 + extra, inner_channel=c["ngf"], out_channel=3, res_blocks=c["res_blocks"],
                     attn_res=c["attn_res"], channel_mults=c["mults"], efficient=c["efficient"])


def build_net(c, dtype, golden_dir, **extra):
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.options import opt_from_json

    opt = opt_from_json({}, overrides_of(c, **extra))
    net = define_G(**vars(opt))
    sched = load(golden_dir, "schedule.pt")
    sd = net.state_dict()
    for k in sd:  # schedule buffers of this implementation == the reference's, bit for bit
        if O._is_buffer(k):
            assert torch.equal(sd[k], sched[k.split(".")[-1]]), k
    syn = O.synth_state_dict(sd, seed=0)
    net.load_state_dict(syn)
    net.jg_finalize(torch.device("cuda:0"), dtype)
    return net, syn


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", CFGS)
def test_unet_vs_reference_golden(golden_dir, name, dtype):
    """UNet forward/backward against the reference's own outputs (tests/golden/unet_*.pt)."""
    from joligen_amd import ops

    g = load(golden_dir, f"unet_{name}.pt")
    net, _ = build_net(g["cfg"], dtype, golden_dir)
    unet = net.denoise_fn.model
    net.arena.ensure_fresh()
    d = torch.device("cuda:0")
    x = ops.to_nhwc(g["x"].to(d), dtype, 8).requires_grad_(True)
    emb = g["emb"].to(d).requires_grad_(True)
    out = unet(x, emb)
    e_out = relerr(ops.to_nchw_f32(out, 3), g["out"])
    assert e_out < TOL_OUT[dtype], ("out", name, e_out)
    R8 = ops.to_nhwc(g["R"].to(d), dtype, 8)
    out.backward(R8)
    torch.cuda.synchronize()
    e_dx = relerr(x.grad.permute(0, 3, 1, 2)[:, :6], g["dx"])
    e_demb = relerr(emb.grad, g["demb"])
    assert e_dx < TOL_GRAD[dtype], ("dx", name, e_dx)
    assert e_demb < TOL_GRAD[dtype], ("demb", name, e_demb)
    bad = []
    for k, ref in g["grad_checks"].items():
        v = dict(unet.named_parameters())[k].grad.detach().float().cpu()
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        tol = TOL_GRAD[dtype] * float(ref[0]) + noise_floor(k, g["grad_checks"], dtype) + 1e-6
        if abs(float(mine[0] - ref[0])) > tol or abs(float(mine[1] - ref[1])) > 2 * tol * max(1.0, v.numel() ** 0.5 / 4):
            bad.append((k, mine.tolist(), ref.tolist(), tol))
    assert not bad, bad[:6]


@pytest.mark.parametrize("name", CFGS)
def test_diffusion_generator_vs_reference_golden(golden_dir, name):
    g = load(golden_dir, f"diffgen_{name}.pt")
    dtype = torch.float16
    net, _ = build_net(g["cfg"], dtype, golden_dir)
    d = torch.device("cuda:0")
    with torch.no_grad():
        noise, noise_hat, w = net(g["B"].to(d), g["A"].to(d), g["mask"].to(d), g["noise"].to(d), t=g["t"], u=g["u"])
    assert torch.equal(noise.cpu(), g["noise"])
    e = relerr(noise_hat, g["noise_hat"])
    assert e < TOL_OUT[dtype], e
    assert relerr(w, g["min_snr_w"]) < 1e-6


class _Opt:
    pass


def make_model(c, dtype_name, golden_dir, hp=None, **extra):
    from joligen_amd.models import create_model
    from joligen_amd.options import opt_from_json

    ov = overrides_of(c, model_type="palette", gpu_ids="0", jg_act_dtype=dtype_name, train_optim="adamw",
                      train_G_ema=True, train_iter_size=1, checkpoints_dir="/tmp/jg_amd_ckpt/", name="t")
    if hp:
        ov.update(train_G_lr=hp["lr"], train_beta1=hp["beta1"], train_beta2=hp["beta2"], train_optim_eps=hp["eps"],
                  train_optim_weight_decay=hp["weight_decay"], train_G_ema_beta=hp["ema_beta"],
                  alg_diffusion_lambda_G=hp["lambda_G"], train_optim=hp["optim"])
    ov.update(extra)
    opt = opt_from_json({}, ov)
    model = create_model(opt, 0)
    sd = model.netG_A.state_dict()
    model.netG_A.load_state_dict(O.synth_state_dict(sd, seed=0))
    model.setup(opt)
    model.single_gpu()
    return model


def palette_trainer(sd, c, hp):
    return O.OraclePaletteTrainer(sd, cfg_of(c), lr=hp["lr"], beta1=hp["beta1"], beta2=hp["beta2"], eps=hp["eps"],
                                  weight_decay=hp["weight_decay"], ema_beta=hp["ema_beta"], lambda_G=hp["lambda_G"], optim=hp["optim"])


# forward-only tolerance of the loss at identical weights, and the minimum cosine between this implementation's single Adam(W)
# update and the oracle's from identical (w, m, v, t).  Adam's first steps are sign-like, so the cosine measures the fraction of
# elements whose 16-bit gradient has the right sign (1 - 2 * flipped): 0 for a missing update, -1 for a mis-signed one.
TOL_LOSS_FWD = {torch.float16: 5e-3, torch.bfloat16: 3e-2}
COS_UPDATE = {torch.float16: 0.90, torch.bfloat16: 0.70}


@pytest.mark.parametrize("dtype_name", ["fp16", "bf16"])
@pytest.mark.parametrize("name", CFGS)
def test_palette_three_steps_vs_reference_golden(golden_dir, name, dtype_name):
    """3 x optimize_parameters() (AdamW + EMA) with the reference's injected (t, u, noise), TEACHER-FORCED (tests/parity_util.py):
    before every iteration the HIP model is given the CPU oracle's state (the oracle itself reproduces the reference's losses and
    parameter checksums of every iteration: tests/test_oracle_golden.py; its loss is re-asserted here), so that each iteration checks a single step: the loss on identical
    weights and the parameter / EMA update against the oracle's update."""
    import parity_util as PU

    g = load(golden_dir, f"palette_step_{name}.pt")
    model = make_model(g["cfg"], dtype_name, golden_dir, g["hp"])
    dtype = torch.float16 if dtype_name == "fp16" else torch.bfloat16
    net = model.netG_A
    sd0 = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tr = palette_trainer(sd0, g["cfg"], g["hp"])
    log = []
    for it, s in enumerate(g["steps"]):
        PU.force_state(net, {k: tr.P[k] for k in tr.param_names}, tr.m, tr.v, tr.step, tr.ema)
        before, ref_before = PU.snapshot(net), {k: tr.P[k].clone() for k in tr.param_names}
        ema_before = None if tr.ema is None else {k: v.clone() for k, v in tr.ema.items()}
        model.rng_injection = lambda b, s=s: (s["t"], s["u"], s["noise"])
        model.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"], "A_img_paths": ["x"]})
        model.optimize_parameters()
        loss = float(model.get_current_losses()["G_tot"])
        loss_ref = float(tr.optimize_parameters(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"]))
        assert abs(loss_ref - float(s["loss"])) < 2e-4 * abs(float(s["loss"])) + 1e-6        # oracle == reference fixture
        assert abs(loss - loss_ref) < TOL_LOSS_FWD[dtype] * abs(loss_ref), (it, loss, loss_ref)
        after = PU.snapshot(net)
        PU.check_update(f"{name} {dtype_name} it{it}", before, after, ref_before, {k: tr.P[k] for k in tr.param_names},
                        COS_UPDATE[dtype], log=log)
        ema = {k: v.detach().float().cpu() for k, v in model.netG_A_ema.named_parameters()}
        PU.check_ema(f"ema it{it}", ema_before, ema, after, g["hp"]["ema_beta"], first=ema_before is None)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/update_agreement_palette_{name}_{dtype_name}.txt", "w") as f:
        f.write("\n".join(log))


@pytest.mark.parametrize("dtype_name", ["fp16", "bf16"])
def test_palette_pix2pix_three_steps_vs_reference_golden(golden_dir, dtype_name):
    """alg_diffusion_task = "pix2pix" (paired conditioning image, no mask: no ground-truth blend, loss over every pixel;
    models/palette_model.py:360-363): the teacher-forced three-step comparison of test_palette_three_steps_vs_reference_golden on the
    fixture of oracle/make_golden_pix2pix.py."""
    import parity_util as PU

    g = load(golden_dir, "palette_step_pix2pix_tiny.pt")
    model = make_model(g["cfg"], dtype_name, golden_dir, g["hp"], alg_diffusion_task="pix2pix")
    dtype = torch.float16 if dtype_name == "fp16" else torch.bfloat16
    net = model.netG_A
    sd0 = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tr = palette_trainer(sd0, g["cfg"], g["hp"])
    for it, s in enumerate(g["steps"]):
        PU.force_state(net, {k: tr.P[k] for k in tr.param_names}, tr.m, tr.v, tr.step, tr.ema)
        before, ref_before = PU.snapshot(net), {k: tr.P[k].clone() for k in tr.param_names}
        model.rng_injection = lambda b, s=s: (s["t"], s["u"], s["noise"])
        model.set_input({"A": s["A"], "B": s["B"], "A_img_paths": ["x"]})
        assert model.mask is None
        model.optimize_parameters()
        loss = float(model.get_current_losses()["G_tot"])
        loss_ref = float(tr.optimize_parameters(s["B"], s["A"], None, s["noise"], s["t"], s["u"]))
        assert abs(loss_ref - float(s["loss"])) < 2e-4 * abs(float(s["loss"])) + 1e-6        # oracle == reference fixture
        assert abs(loss - loss_ref) < TOL_LOSS_FWD[dtype] * abs(loss_ref), (it, loss, loss_ref)
        PU.check_update(f"pix2pix {dtype_name} it{it}", before, PU.snapshot(net), ref_before, {k: tr.P[k] for k in tr.param_names}, COS_UPDATE[dtype])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_unet_attention_heads_of_16_channels_vs_reference_golden(golden_dir, dtype):
    """`G_unet_mha_num_head_channels = 16` (the reference's own run tests): 4 heads of 16 channels -- the unfused attention path (batched
    GEMMs + softmax; the flash kernel serves head dim 32) -- against the reference's UNet output and gradients (tests/golden/unet_heads16.pt)."""
    from joligen_amd import ops

    g = load(golden_dir, "unet_heads16.pt")
    net, _ = build_net(g["cfg"], dtype, golden_dir, G_unet_mha_num_head_channels=g["cfg"]["num_head_channels"])
    unet = net.denoise_fn.model
    net.arena.ensure_fresh()
    d = torch.device("cuda:0")
    x = ops.to_nhwc(g["x"].to(d), dtype, 8).requires_grad_(True)
    emb = g["emb"].to(d).requires_grad_(True)
    assert all(blk.num_heads == 4 for blk in unet.modules() if hasattr(blk, "num_heads") and hasattr(blk, "qkv"))
    out = unet(x, emb)
    e_out = relerr(ops.to_nchw_f32(out, 3), g["out"])
    assert e_out < TOL_OUT[dtype], ("out", e_out)
    out.backward(ops.to_nhwc(g["R"].to(d), dtype, 8))
    torch.cuda.synchronize()
    e_dx = relerr(x.grad.permute(0, 3, 1, 2)[:, :6], g["dx"])
    e_demb = relerr(emb.grad, g["demb"])
    assert e_dx < TOL_GRAD[dtype] and e_demb < TOL_GRAD[dtype], (e_dx, e_demb)
    bad = []
    for k, ref in g["grad_checks"].items():
        v = dict(unet.named_parameters())[k].grad.detach().float().cpu()
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        tol = TOL_GRAD[dtype] * float(ref[0]) + noise_floor(k, g["grad_checks"], dtype) + 1e-6
        if abs(float(mine[0] - ref[0])) > tol or abs(float(mine[1] - ref[1])) > 2 * tol * max(1.0, v.numel() ** 0.5 / 4):
            bad.append((k, mine.tolist(), ref.tolist(), tol))
    assert not bad, bad[:6]


@pytest.mark.parametrize("dtype_name", ["fp16", "bf16"])
def test_palette_minsnr_three_steps_vs_reference_golden(golden_dir, dtype_name):
    """`alg_palette_minsnr = True` (the reference's own run tests select it): teacher-forced three-step comparison on the fixture of
    oracle/make_golden_minsnr.py -- the per-sample weight min(SNR, 5) / SNR enters jg_ddpm_mse_loss and its gradient."""
    import parity_util as PU

    g = load(golden_dir, "palette_step_minsnr_tiny.pt")
    model = make_model(g["cfg"], dtype_name, golden_dir, g["hp"], alg_palette_minsnr=True)
    dtype = torch.float16 if dtype_name == "fp16" else torch.bfloat16
    net = model.netG_A
    hp = g["hp"]
    tr = O.OraclePaletteTrainer({k: v.detach().float().cpu() for k, v in net.state_dict().items()}, cfg_of(g["cfg"]), lr=hp["lr"], beta1=hp["beta1"],
                                beta2=hp["beta2"], eps=hp["eps"], weight_decay=hp["weight_decay"], ema_beta=hp["ema_beta"], lambda_G=hp["lambda_G"],
                                optim=hp["optim"], minsnr=True)
    for it, s in enumerate(g["steps"]):
        PU.force_state(net, {k: tr.P[k] for k in tr.param_names}, tr.m, tr.v, tr.step, tr.ema)
        before, ref_before = PU.snapshot(net), {k: tr.P[k].clone() for k in tr.param_names}
        model.rng_injection = lambda b, s=s: (s["t"], s["u"], s["noise"])
        model.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"], "A_img_paths": ["x"]})
        model.optimize_parameters()
        loss = float(model.get_current_losses()["G_tot"])
        loss_ref = float(tr.optimize_parameters(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"]))
        assert abs(loss_ref - float(s["loss"])) < 2e-4 * abs(float(s["loss"])) + 1e-6
        assert abs(loss - loss_ref) < TOL_LOSS_FWD[dtype] * abs(loss_ref), (it, loss, loss_ref)
        PU.check_update(f"minsnr {dtype_name} it{it}", before, PU.snapshot(net), ref_before, {k: tr.P[k] for k in tr.param_names}, COS_UPDATE[dtype])


@pytest.mark.parametrize("dtype_name", ["fp16", "bf16"])
def test_palette_gradient_accumulation_vs_reference_golden(golden_dir, dtype_name):
    """`train_iter_size = 2` (models/base_model.py:1250-1282,1302-1377) on the fixture of oracle/make_golden_accum.py: per window of two
    calls the HIP model starts from the oracle's state; both calls see identical weights (their losses are forward quantities), the
    parameters must not move on the first call, the optimizer step at the boundary is compared with the oracle's, the EMA recurrence runs
    on every call, and the reported loss is `G_tot_avg` = the window's sum of loss / iter_size."""
    import parity_util as PU

    g = load(golden_dir, "palette_step_accum_tiny.pt")
    n = g["iter_size"]
    model = make_model(g["cfg"], dtype_name, golden_dir, g["hp"], train_iter_size=n)
    dtype = torch.float16 if dtype_name == "fp16" else torch.bfloat16
    net = model.netG_A
    tr = palette_trainer({k: v.detach().float().cpu() for k, v in net.state_dict().items()}, g["cfg"], g["hp"])
    for w0 in range(0, len(g["steps"]), n):
        PU.force_state(net, {k: tr.P[k] for k in tr.param_names}, tr.m, tr.v, tr.step, tr.ema)
        before, ref_before = PU.snapshot(net), {k: tr.P[k].clone() for k in tr.param_names}
        for j in range(n):
            s = g["steps"][w0 + j]
            ema_before = None if tr.ema is None else {k: v.clone() for k, v in tr.ema.items()}
            model.rng_injection = lambda b, s=s: (s["t"], s["u"], s["noise"])
            model.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"], "A_img_paths": ["x"]})
            model.optimize_parameters()
            loss = float(model.loss_G_tot.detach())           # the loss scale of fp16 lives in the gradient only
            loss_ref, reported = tr.iteration(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"], iter_size=n)
            assert abs(float(loss_ref) - float(s["loss_raw"])) < 2e-4 * abs(float(s["loss_raw"])) + 1e-6
            assert abs(loss - float(loss_ref)) < TOL_LOSS_FWD[dtype] * abs(float(loss_ref)), (w0 + j, loss, float(loss_ref))
            after = PU.snapshot(net)
            if j < n - 1:
                assert all(torch.equal(after[k], before[k]) for k in before), "parameters moved inside an accumulation window"
            else:
                PU.check_update(f"accum {dtype_name} window{w0 // n}", before, after, ref_before, {k: tr.P[k] for k in tr.param_names}, COS_UPDATE[dtype])
                avg = float(model.get_current_losses()["G_tot_avg"])
                assert abs(avg - float(reported)) < TOL_LOSS_FWD[dtype] * abs(float(reported)), (avg, float(reported))
            ema = {k: v.detach().float().cpu() for k, v in model.netG_A_ema.named_parameters()}
            PU.check_ema(f"accum ema it{w0 + j}", ema_before, ema, after, g["hp"]["ema_beta"], first=ema_before is None)


@pytest.mark.parametrize("dtype_name", ["fp16"])
def test_first_step_gradients_vs_oracle_medium(golden_dir, dtype_name):
    """A larger seeded case than the fixtures (64x64, B=2, ngf 32, 3 levels) against the CPU oracle:
    loss, noise_hat and every parameter gradient of the first step."""
    c = dict(ngf=32, mults=[1, 2, 4], res_blocks=[2, 2, 1], attn_res=[16], efficient=True, S=64, B=2)
    model = make_model(c, dtype_name, golden_dir, train_G_ema=False)
    net = model.netG_A
    g = torch.Generator().manual_seed(5)
    Bimg = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    mask = torch.zeros(2, 1, 64, 64, dtype=torch.int64)
    mask[:, :, 10:40, 20:50] = 1
    A = Bimg * (1 - mask) + torch.randn(2, 3, 64, 64, generator=g) * mask
    t, u, noise = O.draw_step_randomness(torch.Generator().manual_seed(9), Bimg, 2000)
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tr = O.OraclePaletteTrainer(sd, cfg_of(c), ema_beta=None)
    loss_ref, grads_ref, nh_ref = tr.loss_and_grads(Bimg, A, mask, noise, t, u)
    model.rng_injection = lambda b: (t, u, noise)
    model.set_input({"A": A, "B": Bimg, "B_label_mask": mask})
    model.compute_palette_loss()
    model.loss_G_tot.backward()
    torch.cuda.synchronize()
    assert abs(float(model.loss_G_tot) - float(loss_ref)) < 5e-3 * float(loss_ref)
    scale = model.loss_scale
    dtype = torch.float16
    ref_norms = {k: float(v.norm()) for k, v in grads_ref.items()}
    worst, table = [], []
    for k, p in net.named_parameters():
        gr = grads_ref[k]
        mine = (p.grad / scale).detach().float().cpu()
        floor = noise_floor(k, ref_norms, dtype)
        e = float((mine - gr).norm() / (gr.norm() + floor + 1e-12))
        worst.append((e, k))
        table.append(f"{e:10.3e} ref={float(gr.norm()):10.3e} mine={float(mine.norm()):10.3e} floor={floor:9.2e} {k}")
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/grad_table_medium.txt", "w") as f:
        f.write("\n".join(table))
    worst.sort(reverse=True)
    assert worst[0][0] < 6e-2, worst[:8]
    med = sorted(e for e, _ in worst)[len(worst) // 2]
    assert med < 1.5e-2, med


def _first_step_vs_oracle(c, dtype_name, golden_dir, tag, seed=5, with_floor=False):
    """loss, noise_hat and every parameter gradient of the first step of configuration `c` against the CPU oracle;
    returns [(relative error, name)] sorted worst first and writes the table to gpurun_out/grad_table_<tag>.txt.
    with_floor: also the rounding floor of THIS case, measured on the spot (the oracle with 16-bit storage of activations, activation
    gradients and convolution weights, `activation_rounding`): returned as the sorted list of its per-parameter errors."""
    model = make_model(c, dtype_name, golden_dir, train_G_ema=False)
    net = model.netG_A
    B, S = c["B"], c["S"]
    g = torch.Generator().manual_seed(seed)
    Bimg = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    mask = torch.zeros(B, 1, S, S, dtype=torch.int64)
    mask[:, :, S // 6:(5 * S) // 8, S // 3:(7 * S) // 9] = 1
    A = Bimg * (1 - mask) + torch.randn(B, 3, S, S, generator=g) * mask
    t, u, noise = O.draw_step_randomness(torch.Generator().manual_seed(9), Bimg, 2000)
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tr = O.OraclePaletteTrainer(sd, cfg_of(c), ema_beta=None)
    loss_ref, grads_ref, nh_ref = tr.loss_and_grads(Bimg, A, mask, noise, t, u)
    model.rng_injection = lambda b: (t, u, noise)
    model.set_input({"A": A, "B": Bimg, "B_label_mask": mask})
    model.compute_palette_loss()
    model.loss_G_tot.backward()
    torch.cuda.synchronize()
    dtype = torch.float16 if dtype_name == "fp16" else torch.bfloat16
    loss = float(model.loss_G_tot)
    scale = model.loss_scale
    ref_norms = {k: float(v.norm()) for k, v in grads_ref.items()}
    worst, table = [], []
    for k, p in net.named_parameters():
        gr = grads_ref[k]
        mine = (p.grad / scale).detach().float().cpu()
        floor = noise_floor(k, ref_norms, dtype)
        e = float((mine - gr).norm() / (gr.norm() + floor + 1e-12))
        worst.append((e, k))
        table.append(f"{e:10.3e} ref={float(gr.norm()):10.3e} mine={float(mine.norm()):10.3e} floor={floor:9.2e} {k}")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/grad_table_{tag}.txt", "w") as f:
        f.write(f"loss {loss:.6f} oracle {float(loss_ref):.6f}\n" + "\n".join(table))
    worst.sort(reverse=True)
    if with_floor:
        sd16 = {k: (v.to(dtype).float() if (torch.is_floating_point(v) and v.dim() >= 3) else v) for k, v in sd.items()}
        tr16 = O.OraclePaletteTrainer(sd16, cfg_of(c), ema_beta=None)
        tr16.grad_scale = scale
        with O.activation_rounding(dtype):
            _, grads16, _ = tr16.loss_and_grads(Bimg, A, mask, noise, t, u)
        fl = sorted((float((grads16[k] - grads_ref[k]).norm() / (grads_ref[k].norm() + noise_floor(k, ref_norms, dtype) + 1e-12)) for k in grads_ref), reverse=True)
        with open(f"gpurun_out/grad_table_{tag}.txt", "a") as f:
            f.write(f"\n# rounding floor of this case: worst {fl[0]:.3e} median {fl[len(fl) // 2]:.3e} | HIP worst {worst[0][0]:.3e} median {sorted(e for e, _ in worst)[len(worst) // 2]:.3e}")
        return loss, float(loss_ref), worst, fl
    return loss, float(loss_ref), worst


# BASELINE configs[1] (C2) and configs[3] (C4) architectures at their full image size, batch 1 (what the CPU oracle does in seconds):
# ngf 64, mults [1,2,4,8], two ResBlocks per level, 1024-channel skip concats, mid-block attention at T = 1024 / 4096, the split-K
# weight gradients of the full-resolution layers, the sub-pixel / upsample-on-read / pooled-store modes of the up path.
C2 = dict(ngf=64, mults=[1, 2, 4, 8], res_blocks=[2, 2, 2, 2], attn_res=[16], efficient=True, S=256, B=1)
C2_NOEFF = dict(C2, efficient=False)
C4 = dict(C2, efficient=False, S=512)


@pytest.mark.parametrize("cname,c,dtype_name", [("c2_eff_fp16", C2, "fp16"), ("c2_eff_bf16", C2, "bf16"), ("c2_noeff_fp16", C2_NOEFF, "fp16"),
                                                ("c4_512_fp16", C4, "fp16")])
def test_first_step_gradients_vs_oracle_baseline_shapes(golden_dir, cname, c, dtype_name):
    loss, loss_ref, worst = _first_step_vs_oracle(c, dtype_name, golden_dir, cname)
    fp16 = dtype_name == "fp16"
    assert abs(loss - loss_ref) < (5e-3 if fp16 else 3e-2) * loss_ref, (loss, loss_ref)
    med = sorted(e for e, _ in worst)[len(worst) // 2]
    assert worst[0][0] < (8e-2 if fp16 else 0.4), worst[:8]
    assert med < (1.5e-2 if fp16 else 8e-2), med
    if cname.startswith("c2_eff"):
        # the MEASURED rounding floor of this exact case (fp32 oracle with 16-bit storage of activations, activation gradients and conv
        # weights: tests/test_oracle_golden.py::test_rounding_yardstick -> profiles/r02_rounding_yardstick.json): the HIP step has to
        # sit on it, not merely under a tolerance argued from layer counts
        import json
        y = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_rounding_yardstick.json")))
        y = y["c2_256"][dtype_name]
        assert med < 1.5 * y["grad_median"], (med, y)
        assert worst[0][0] < 2.0 * y["grad_worst"], (worst[:4], y)
        assert abs(loss - loss_ref) / loss_ref < max(4.0 * y["loss_rel"], 3e-4), (loss, loss_ref, y)


@pytest.mark.parametrize("cname,c", [("c2_noeff_bf16", C2_NOEFF), ("c4_512_bf16", C4)])
def test_first_step_gradients_bf16_vs_measured_floor(golden_dir, cname, c):
    """bf16 -- the bench dtype -- on the non-efficient C2 and the 512x512 C4 architectures: every gradient against the
    CPU oracle, bounded by the rounding floor of the same case measured on the spot instead of a loose fixed tolerance"""
    loss, loss_ref, worst, fl = _first_step_vs_oracle(c, "bf16", golden_dir, cname, with_floor=True)
    assert abs(loss - loss_ref) < 3e-2 * loss_ref, (loss, loss_ref)
    med = sorted(e for e, _ in worst)[len(worst) // 2]
    assert worst[0][0] <= 2.0 * fl[0], (worst[:4], fl[:3])
    assert med <= 1.5 * fl[len(fl) // 2], (med, fl[len(fl) // 2])


@pytest.mark.parametrize("lossname", ["L1", "multiscale_L1", "multiscale_MSE"])
def test_palette_loss_variants_first_step_vs_oracle(golden_dir, lossname):
    """alg_palette_loss != MSE through the model API (64x64 so that two resolutions exist): total loss, the logged per-resolution
    terms and the parameter gradients of the first step against the CPU oracle."""
    c = dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[16], efficient=True, S=64, B=2)
    model = make_model(c, "fp16", golden_dir, train_G_ema=False, alg_palette_loss=lossname, alg_diffusion_lambda_G=2.0)
    net = model.netG_A
    g = torch.Generator().manual_seed(6)
    Bimg = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    mask = torch.zeros(2, 1, 64, 64, dtype=torch.int64)
    mask[:, :, 8:44, 16:56] = 1
    A = Bimg * (1 - mask) + torch.randn(2, 3, 64, 64, generator=g) * mask
    t, u, noise = O.draw_step_randomness(torch.Generator().manual_seed(10), Bimg, 2000)
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tr = O.OraclePaletteTrainer(sd, cfg_of(c), ema_beta=None, lambda_G=2.0, lossname=lossname)
    loss_ref, grads_ref, nh_ref = tr.loss_and_grads(Bimg, A, mask, noise, t, u)
    model.rng_injection = lambda b: (t, u, noise)
    model.set_input({"A": A, "B": Bimg, "B_label_mask": mask})
    model.compute_palette_loss()
    model.loss_G_tot.backward()
    torch.cuda.synchronize()
    assert abs(float(model.loss_G_tot) - float(loss_ref)) < 5e-3 * float(loss_ref), (float(model.loss_G_tot), float(loss_ref))
    if lossname.startswith("multiscale"):
        _, lev = O.palette_loss_variants(noise, nh_ref, mask, lossname)
        assert model.loss_names_G == ["G_tot", "G_32", "G_64", "G_64"]
        for k, v in lev.items():
            assert abs(float(getattr(model, "loss_G_" + k)) - float(v)) < 5e-3 * float(v), k
    ref_norms = {k: float(v.norm()) for k, v in grads_ref.items()}
    errs = []
    for k, p in net.named_parameters():
        gr = grads_ref[k]
        mine = (p.grad / model.loss_scale).detach().float().cpu()
        errs.append((float((mine - gr).norm() / (gr.norm() + noise_floor(k, ref_norms, torch.float16) + 1e-12)), k))
    errs.sort(reverse=True)
    # L1's sign() gradient flips wherever the 16-bit prediction error crosses zero: looser than the MSE bound
    assert errs[0][0] < (0.2 if lossname.endswith("L1") else 6e-2), errs[:6]
    assert sorted(e for e, _ in errs)[len(errs) // 2] < (6e-2 if lossname.endswith("L1") else 1.5e-2)


def test_checkpoint_roundtrip_reference_layout(golden_dir, tmp_path):
    """save_networks writes `<suffix>_net_G_A.pth` with the reference's keys, plain contiguous
    fp32 tensors in the reference's logical (OIHW) layout, loadable by both sides."""
    g = load(golden_dir, "palette_step_tiny_eff.pt")
    model = make_model(g["cfg"], "bf16", golden_dir, checkpoints_dir=str(tmp_path) + "/", name="ck")
    s = g["steps"][0]
    model.rng_injection = lambda b: (s["t"], s["u"], s["noise"])
    model.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"]})
    model.optimize_parameters()
    model.save_networks("latest")
    sd = torch.load(os.path.join(str(tmp_path), "ck", "latest_net_G_A.pth"), map_location="cpu")
    assert list(sd.keys()) == g["keys"]
    for k, v in sd.items():
        assert tuple(v.shape) == g["shapes"][k] and v.is_contiguous() and v.dtype in (torch.float32,), k
    ema = torch.load(os.path.join(str(tmp_path), "ck", "latest_net_G_A_ema.pth"), map_location="cpu")
    assert list(ema.keys()) == g["keys"]
    model2 = make_model(g["cfg"], "bf16", golden_dir, checkpoints_dir=str(tmp_path) + "/", name="ck")
    model2.load_networks("latest")
    for (k, a), (_, b) in zip(model.netG_A.state_dict().items(), model2.netG_A.state_dict().items()):
        assert torch.equal(a, b), k
    # the oracle consumes the checkpoint as is
    tr = O.OraclePaletteTrainer(sd, cfg_of(g["cfg"]))
    assert set(tr.P) == set(g["keys"])
    # asynchronous writer: the step loop pays one device -> pinned copy per arena; the files written by the
    # background thread are bit-identical to the synchronous ones, and a training step launched right after the save (it changes
    # the weights on the device) does not leak into the snapshot
    model.save_networks("async", blocking=False)
    model.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"]})
    model.optimize_parameters()
    model.wait_checkpoints()
    for suffix in ("G_A", "G_A_ema"):
        a = torch.load(os.path.join(str(tmp_path), "ck", f"async_net_{suffix}.pth"), map_location="cpu")
        b = torch.load(os.path.join(str(tmp_path), "ck", f"latest_net_{suffix}.pth"), map_location="cpu")
        assert list(a.keys()) == list(b.keys()) == g["keys"]
        for k in a:
            assert torch.equal(a[k], b[k]) and a[k].is_contiguous(), (suffix, k)
    model.export_networks("latest")       # palette: nothing to export, like the reference


def test_train_continue_from_loads_source_run(golden_dir, tmp_path):
    """The device half of /root/reference/tests/test_train_continue_from.py (`test_train_continue_from_loads_weights_from_source_dir`,
    `..._uses_train_load_iter_suffix`) on a real model: `setup()` with `train_continue_from=<dir>` loads `<dir>/<suffix>_net_G_A.pth`
    (suffix = `latest`, or `iter_<n>` with `train_load_iter`), and the run keeps saving under ITS OWN `<checkpoints_dir>/<name>`."""
    g = load(golden_dir, "palette_step_tiny_eff.pt")
    ck = str(tmp_path) + "/"
    src = make_model(g["cfg"], "bf16", golden_dir, checkpoints_dir=ck, name="source_run")

    base = {k: v.clone() for k, v in src.netG_A.state_dict().items()}
    src.netG_A.load_state_dict({k: (v + 0.25 if v.is_floating_point() and not O._is_buffer(k) else v) for k, v in base.items()})
    src.save_networks("latest")
    src.netG_A.load_state_dict({k: (v + 0.5 if v.is_floating_point() and not O._is_buffer(k) else v) for k, v in base.items()})
    src.save_networks("iter_123")
    source_dir = os.path.join(ck, "source_run")
    for extra, suffix in ((dict(), "latest"), (dict(train_load_iter=123), "iter_123")):
        tgt = make_model(g["cfg"], "bf16", golden_dir, checkpoints_dir=ck, name="target_run", train_continue_from=source_dir, **extra)
        assert tgt.save_dir == os.path.join(ck, "target_run")
        want = torch.load(os.path.join(source_dir, f"{suffix}_net_G_A.pth"), map_location="cpu")
        got = tgt.netG_A.state_dict()
        moved = 0
        for k in want:
            assert torch.equal(got[k].cpu(), want[k]), (suffix, k)
            moved += int(not torch.equal(want[k], base[k].cpu()))
        assert moved > 100            # the checkpoint really differs from the weights the model was built with


def test_full_size_properties():
    """BASELINE config-2 layer sizes (B=4 instead of 32 to bound test time): properties that do not
    need the CPU oracle -- linearity of the conv kernels and agreement of the MFMA conv with
    PyTorch's own fp32 GPU convolution at 256x256."""
    import torch.nn.functional as F

    from joligen_amd import ops

    d = torch.device("cuda:0")
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    for (B, S, Cin, Cout) in [(4, 256, 64, 64), (4, 128, 128, 128), (4, 32, 512, 512)]:
        x = torch.randn(B, S, S, Cin, generator=g).to(dtype).to(d)
        w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (3 * Cin ** 0.5)).to(dtype).to(d)
        y = torch.ops.jg355.conv2d_nt(x, w, None, None, 1, 1, 1.0, 0.0)
        y2 = torch.ops.jg355.conv2d_nt(x, w, None, None, 1, 1, 2.0, 0.0)
        assert relerr(y2.float(), 2 * y.float()) < 4e-3          # linearity in alpha
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        assert relerr(y.float(), ref) < 8e-3, (B, S, Cin, Cout)


def test_resblock_dropout_in_training(golden_dir):
    """`nn.Dropout(p)` of ResBlock.out_layers (reference unet_generator_attn.py:207-215, 262): between SiLU and the second convolution,
    Bernoulli(1 - p) mask scaled by 1 / (1 - p), training mode only.  Checked against the same network with dropout 0 and the mask applied
    from outside (a forward pre-hook on the second convolution drawing the SAME uniforms): same output and input gradient;
    eval mode ignores it; the default source zeroes a fraction p."""
    from joligen_amd import ops
    from joligen_amd.modules.unet_generator_attn import ResBlock

    p_drop = 0.25
    c = dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[], efficient=False, S=32, B=2)
    net, _ = build_net(c, torch.float16, golden_dir)
    unet = net.denoise_fn.model
    net.arena.ensure_fresh()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x0 = ops.to_nhwc(torch.randn(c["B"], 6, c["S"], c["S"], generator=g).to(d), torch.float16, 8)
    emb0 = torch.randn(c["B"], unet.cond_embed_dim, generator=g).to(d)
    R = ops.to_nhwc(torch.randn(c["B"], 3, c["S"], c["S"], generator=g).to(d), torch.float16, 8)
    blocks = [m for m in unet.modules() if isinstance(m, ResBlock)]
    assert len(blocks) >= 5
    gen = torch.Generator(device=d)

    def src(shape, device):
        return torch.rand(shape, generator=gen, device=device)

    def set_drop(p, source):
        unet.dropout = p
        for rb in blocks:
            rb.dropout, rb.dropout_rand = p, source

    def run(train=True):
        unet.train(train)
        net.arena.g.zero_()
        x = x0.clone().requires_grad_(True)
        out = unet(x, emb0.clone())
        out.backward(R)
        torch.cuda.synchronize()
        return out.detach().clone(), x.grad.detach().clone()

    try:
        set_drop(0.0, None)
        base_out, base_dx = run()
        eval_out, _ = run(train=False)
        # (a) dropout inside the blocks, injected uniforms
        set_drop(p_drop, src)
        gen.manual_seed(11)
        out_a, dx_a = run()
        # (b) dropout 0, the same mask applied in front of the second convolution from outside
        set_drop(0.0, None)
        unet.jg_fused = False
        zero_frac = []

        def hook(mod, args, kwargs):
            h = args[0]
            u = src(h.shape, h.device)
            zero_frac.append(float((u >= 1.0 - p_drop).float().mean()))
            return (h * ((u < 1.0 - p_drop).to(h.dtype) * (1.0 / (1.0 - p_drop))),) + tuple(args[1:]), kwargs

        hs = [rb.out_layers[3].register_forward_pre_hook(hook, with_kwargs=True) for rb in blocks]
        gen.manual_seed(11)
        out_b, dx_b = run()
        for h_ in hs:
            h_.remove()
        unet.jg_fused = True
        # (not bit-identical: the GroupNorm statistics are fp32 atomics, their summation order differs from run to run)
        assert relerr(out_a, out_b) < 2e-3 and relerr(dx_a, dx_b) < 1e-2, (relerr(out_a, out_b), relerr(dx_a, dx_b))
        assert relerr(out_a, base_out) > 0.05          # it did something
        assert abs(sum(zero_frac) / len(zero_frac) - p_drop) < 0.02
        # (c) eval mode: the mask is off, the fused schedule runs
        set_drop(p_drop, None)
        eval_drop, _ = run(train=False)
        assert relerr(eval_drop, eval_out) < 2e-3, relerr(eval_drop, eval_out)
        # (d) default source: runs, finite, differs from the undropped output
        out_d, dx_d = run()
        assert torch.isfinite(out_d.float()).all() and torch.isfinite(dx_d.float()).all() and relerr(out_d, base_out) > 0.05
    finally:
        set_drop(0.0, None)
        unet.jg_fused = True
        unet.train(True)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("efficient", [True, False])
def test_fused_schedule_matches_module_graph(golden_dir, efficient, dtype):
    """unet_exec.py (one autograd node: conv-epilogue statistics, concat-free skip connections, fused
    gradient fan-in, halo-resident 3x3 kernels) against the module-by-module autograd graph of the
    same network at a size where every fused path is live (C % 64 == 0, H % 16 == 0)."""
    from joligen_amd import ops

    c = dict(ngf=64, mults=[1, 2], res_blocks=[1, 1], attn_res=[2], efficient=efficient, S=64, B=2)
    net, _ = build_net(c, dtype, golden_dir)
    unet = net.denoise_fn.model
    net.arena.ensure_fresh()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    x0 = ops.to_nhwc(torch.randn(c["B"], 6, c["S"], c["S"], generator=g).to(d), dtype, 8)
    emb0 = torch.randn(c["B"], unet.cond_embed_dim, generator=g).to(d)
    R = ops.to_nhwc(torch.randn(c["B"], 3, c["S"], c["S"], generator=g).to(d), dtype, 8)
    from joligen_amd.modules import unet_exec

    res = {}
    keep = unet_exec.FUSE_GN_REDUCE          # (off by default, DESIGN.md 13: the flag must not leak into the tests that follow)
    try:
        for fused in (False, True, "no_reduce_fusion"):
            unet.jg_fused = bool(fused)
            unet_exec.FUSE_GN_REDUCE = fused is True
            net.arena.g.zero_()
            x = x0.clone().requires_grad_(True)
            emb = emb0.clone().requires_grad_(True)
            out = unet(x, emb)
            out.backward(R)
            torch.cuda.synchronize()
            res[fused] = (out.detach().float(), x.grad.float(), emb.grad.clone(), net.arena.g.clone())
    finally:
        unet_exec.FUSE_GN_REDUCE = keep
    # GroupNorm-backward reductions in the dgrad epilogue vs the separate reduction pass: same sums
    assert relerr(res[True][3], res["no_reduce_fusion"][3]) < TOL_OUT[dtype], relerr(res[True][3], res["no_reduce_fusion"][3])
    tol = 4 * TOL_OUT[dtype]
    assert relerr(res[True][0], res[False][0]) < tol, ("out", relerr(res[True][0], res[False][0]))
    assert relerr(res[True][1], res[False][1]) < 2 * tol, ("dx", relerr(res[True][1], res[False][1]))
    assert relerr(res[True][2], res[False][2]) < 2 * tol, ("demb", relerr(res[True][2], res[False][2]))
    assert relerr(res[True][3], res[False][3]) < 2 * tol, ("flat grad", relerr(res[True][3], res[False][3]))
    bad = []
    for name, p in unet.named_parameters():
        if name.endswith(".weight") and p.dim() == 4:
            off, n = net.arena.slices["denoise_fn.model." + name]
            a, b = res[True][3][off:off + n], res[False][3][off:off + n]
            if float(b.norm()) > 0 and relerr(a, b) > 2 * tol:
                bad.append((name, relerr(a, b)))
    assert not bad, bad[:8]


_RCCL_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, os.path.join({root!r}, "tests"))
import jg_oracle as O
from joligen_amd import parallel
from joligen_amd.models import create_model
from joligen_amd.options import opt_from_json

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29731")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
c = dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[16], efficient=True, S=32, B=2)
ov = dict(G_ngf=32, G_unet_mha_channel_mults=[1, 2], G_unet_mha_res_blocks=[1, 1], G_unet_mha_attn_res=[16],
          G_unet_mha_vit_efficient=True, data_crop_size=32, train_batch_size=2, model_type="palette", gpu_ids="0",
          jg_act_dtype="fp16", train_optim="adamw", train_G_ema=True, train_iter_size=1, checkpoints_dir="/tmp/jg_rccl/", name="t")
g = torch.Generator().manual_seed(5)
Bimg = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
mask = torch.zeros(2, 1, 32, 32, dtype=torch.int64); mask[:, :, 8:24, 4:20] = 1
A = Bimg * (1 - mask) + torch.randn(Bimg.shape, generator=g) * mask
draws = [O.draw_step_randomness(torch.Generator().manual_seed(10 + i), Bimg, 2000) for i in range(3)]
out = []
for mode in ("single", "rccl"):
    opt = opt_from_json({{}}, ov)
    model = create_model(opt, 0)
    model.netG_A.load_state_dict(O.synth_state_dict(model.netG_A.state_dict(), seed=0))
    model.setup(opt)
    if mode == "single":
        model.single_gpu()
    else:
        parallel.FORCE_EXCHANGE = True
        model.parallelize(0)          # broadcast + FlatDataParallel wrapper
    losses = []
    for d in draws:
        model.rng_injection = lambda b, d=d: d
        model.set_input({{"A": A, "B": Bimg, "B_label_mask": mask}})
        model.optimize_parameters()
        losses.append(float(model.get_current_losses()["G_tot"]))
    net = model.netG_A.module if hasattr(model.netG_A, "module") else model.netG_A
    out.append((losses, float(net.arena.p.double().norm()), float(net.arena.ema.double().norm())))
    if mode == "rccl":   # the reduction of most chunks was started from inside the backward (parallel.EarlyExchange)
        ex = net.arena.early_exchange
        assert ex.last_early >= len(ex.bounds) - 2, (ex.last_early, len(ex.bounds), ex.total)
    parallel.FORCE_EXCHANGE = False
dist.destroy_process_group()
(l0, p0, e0), (l1, p1, e1) = out
# fp32 atomics (split-K, statistics) make two runs differ at the 1e-5 level even in the same mode
assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(l0, l1)), (l0, l1)
assert abs(p0 - p1) <= 1e-4 * p0 and abs(e0 - e1) <= 1e-4 * e0, (p0, p1, e0, e1)
print("RCCL_PATH_OK", l1)
"""


def test_rccl_exchange_path_single_rank():
    """The multi-GPU code path on the one GPU of the test box: a 1-rank RCCL process group, parameters broadcast,
    FlatDataParallel wrapper, chunked async all-reduce of the flat gradient with the fused optimizer pipelined
    behind it -- three steps must reproduce the single-GPU path (sum over one rank, mean over one rank)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT.format(root=root)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "RCCL_PATH_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name", ["tiny_eff", "tiny_attn"])
def test_ddpm_restoration_vs_reference_golden(golden_dir, name, dtype):
    """DiffusionGenerator.restoration (DDPM sampler, short test schedule) against the reference's own samples with
    the per-step noise injected: one UNet forward + one fused `jg_ddpm_p_sample` kernel per step."""
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.options import opt_from_json

    g = load(golden_dir, f"sampling_{name}.pt")
    opt = opt_from_json({}, overrides_of(g["cfg"], G_diff_n_timestep_test=g["T"]))
    net = define_G(**vars(opt))
    sd = net.state_dict()
    for k, v in g["sched_test"].items():   # schedule buffers of this implementation == the reference's, bit for bit
        assert torch.equal(sd["denoise_fn.model." + k], v), k
    net.load_state_dict(O.synth_state_dict(sd, seed=0))
    net.jg_finalize(torch.device("cuda:0"), dtype)
    d = torch.device("cuda:0")
    y, ret = net.restoration(g["A"].to(d), y_t=g["y_t0"].to(d), y_0=g["B"].to(d), mask=g["mask"].to(d), sample_num=2,
                             noises=g["noises"])
    torch.cuda.synchronize()
    assert ret.shape == g["ret"].shape
    # unmasked pixels are exact copies of y_0 at every recorded step (mask blend is bit exact)
    keep = (g["mask"] == 0).expand_as(g["B"])
    assert torch.equal(y.cpu()[keep], g["B"][keep])
    e = relerr(y, g["y_out"])
    assert e < 2 * TOL_OUT[dtype], e
    assert relerr(ret, g["ret"]) < 2 * TOL_OUT[dtype]
    # DDIM sampler: same kernel, other coefficients
    net.set_new_sampling_method("ddim")
    y, ret = net.restoration(g["A"].to(d), y_t=g["y_t0"].to(d), y_0=g["B"].to(d), mask=g["mask"].to(d), sample_num=2,
                             ddim_num_steps=4, ddim_eta=0.5)
    assert ret.shape == g["ret_ddim"].shape
    assert torch.equal(y.cpu()[keep], g["B"][keep])
    assert relerr(y, g["y_ddim"]) < 2 * TOL_OUT[dtype], relerr(y, g["y_ddim"])
    assert relerr(ret, g["ret_ddim"]) < 2 * TOL_OUT[dtype]


def test_fp16_overflow_drops_step_and_backs_off_loss_scale(golden_dir):
    """the fp16 path used a static loss scale with no overflow detection.  An absurd loss scale makes the first backward
    overflow: the fused optimizer drops those steps on the device (weights, moments, EMA stay finite and unchanged), the poll halves
    the scale until steps go through, and the dropped steps do not advance the bias correction."""
    g = load(golden_dir, "palette_step_tiny_eff.pt")
    model = make_model(g["cfg"], "fp16", golden_dir, g["hp"], jg_loss_scale=2.0 ** 40, jg_overflow_poll=1)
    net = model.netG_A
    w0 = net.arena.p.clone()
    s = g["steps"][0]
    model.rng_injection = lambda b: (s["t"], s["u"], s["noise"])
    scales = []
    for it in range(40):
        model.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"], "A_img_paths": ["x"]})
        model.optimize_parameters()
        scales.append(model.loss_scale)
        if it == 0:
            assert torch.equal(net.arena.p, w0) and net.arena.step == 0, "an overflowing step must leave the weights alone"
            assert int(net.arena.overflow[1]) == 1
    assert torch.isfinite(net.arena.p).all() and torch.isfinite(net.arena.m).all() and torch.isfinite(net.arena.v).all()
    assert scales[-1] < scales[0] and not torch.equal(net.arena.p, w0), scales[-5:]
    assert net.arena.step == 40 - int(net.arena.overflow[1])
    # and radam / lion run through the model API
    for name in ("radam", "lion"):
        m2 = make_model(g["cfg"], "bf16", golden_dir, train_optim=name)
        m2.rng_injection = lambda b: (s["t"], s["u"], s["noise"])
        before = m2.netG_A.arena.p.clone()
        m2.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"], "A_img_paths": ["x"]})
        m2.optimize_parameters()
        assert not torch.equal(m2.netG_A.arena.p, before) and torch.isfinite(m2.netG_A.arena.p).all()


def test_palette_compute_visuals_surface(golden_dir):
    """the display path of train.py:319-413 through the model API: compute_visuals(n) runs the sampler on the
    current batch (palette_model.py:622-887), get_current_visuals(n) returns one OrderedDict per image with the reference's names;
    the sampled output equals a direct restoration() call with the same draws and keeps the unmasked pixels bit-exact."""
    g = load(golden_dir, "sampling_tiny_eff.pt")
    c = g["cfg"]
    model = make_model(c, "fp16", golden_dir, G_diff_n_timestep_test=g["T"])
    B = c["B"]
    model.set_input({"A": g["y_t0"], "B": g["B"], "B_label_mask": g["mask"], "A_img_paths": ["x"] * B})
    model.sampling_noises = g["noises"]
    model.compute_visuals(B)
    vis = model.get_current_visuals(B)
    assert len(vis) == B
    assert list(vis[0].keys()) == ["gt_image_0", "cond_image_0", "y_t_0", "mask_0", "output_0"]
    out = torch.cat([v[f"output_{k}"] for k, v in enumerate(vis)])
    d = torch.device("cuda:0")
    y, _ = model.netG_A.restoration(g["y_t0"].to(d), y_t=g["y_t0"].to(d), y_0=g["B"].to(d), mask=g["mask"].to(d), sample_num=2, noises=g["noises"])
    assert relerr(out, y) < 2 * TOL_OUT[torch.float16], relerr(out, y)     # two runs differ by the order of the statistics atomics
    keep = (g["mask"] == 0).expand_as(g["B"])
    assert torch.equal(out.cpu()[keep], g["B"][keep])
    assert model.fake_B is model.output and tuple(vis[1]["mask_1"].shape) == tuple(g["mask"].shape[1:])


@pytest.mark.parametrize("dtype_name", ["bf16", "fp16"])
def test_loss_curve_200_steps_vs_oracle(golden_dir, dtype_name):
    """north_star: "loss curves matching the CPU reference within tolerance".  200 free-running optimize_parameters() steps
    (AdamW + EMA, lr 2e-4) of a 3-level UNet (ngf 32, 64x64, batch 2) on HIP against the CPU oracle trainer from the same weights,
    the same fresh synthetic batch and the same injected (t, u, noise) at every step -- NOT teacher-forced: the two weight
    trajectories are free to separate, what must agree is the training signal.  Compared: the curves smoothed over 20 steps
    (relative band), the mean of the last 50 steps, and that both actually learn.  The raw curves go to gpurun_out/ (committed under
    profiles/)."""
    import json

    c = dict(ngf=32, mults=[1, 2, 4], res_blocks=[1, 1, 1], attn_res=[16], efficient=True, S=64, B=2)
    n_steps = 200
    model = make_model(c, dtype_name, golden_dir, train_G_ema=True)
    net = model.netG_A
    # the reference's zero_module() layers start at zero and a DDPM loss starts at 1: keep the deterministic synthetic weights but zero
    # the layers the reference zeroes, so that the curve has the reference's shape (a descent from ~1)
    sd = net.state_dict()
    for k in sd:
        if k.endswith("out_layers.3.weight") or k.endswith("out_layers.3.bias") or ".proj_out." in k or k.endswith("model.out.2.weight") or k.endswith("model.out.2.bias"):
            sd[k] = torch.zeros_like(sd[k])
    net.load_state_dict(sd)
    net.arena.refresh()
    sd0 = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tr = O.OraclePaletteTrainer(sd0, cfg_of(c), lr=model.opt.train_G_lr, ema_beta=model.opt.train_G_ema_beta)
    gd, gr = torch.Generator().manual_seed(11), torch.Generator().manual_seed(12)
    mine, ref = [], []
    for it in range(n_steps):
        Bimg = torch.rand(c["B"], 3, c["S"], c["S"], generator=gd) * 2 - 1
        mask = torch.zeros(c["B"], 1, c["S"], c["S"], dtype=torch.int64)
        h0, w0 = int(torch.randint(0, 24, (1,), generator=gd)), int(torch.randint(0, 24, (1,), generator=gd))
        mask[:, :, h0:h0 + 32, w0:w0 + 36] = 1
        A = Bimg * (1 - mask) + torch.randn(Bimg.shape, generator=gd) * mask
        t, u, noise = O.draw_step_randomness(gr, Bimg, 2000)
        model.rng_injection = lambda b, t=t, u=u, noise=noise: (t, u, noise)
        model.set_input({"A": A, "B": Bimg, "B_label_mask": mask})
        model.optimize_parameters()
        mine.append(float(model.get_current_losses()["G_tot"]))
        ref.append(float(tr.optimize_parameters(Bimg, A, mask, noise, t, u)))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/loss_curve_200_{dtype_name}.json", "w") as f:
        json.dump({"config": c, "dtype": dtype_name, "steps": n_steps, "hip": mine, "cpu_oracle": ref}, f)
    m, r = torch.tensor(mine), torch.tensor(ref)
    assert torch.isfinite(m).all()
    win = 20
    ms, rs = m.unfold(0, win, 1).mean(1), r.unfold(0, win, 1).mean(1)
    band = float(((ms - rs).abs() / rs).max())
    tail = abs(float(m[-50:].mean()) - float(r[-50:].mean())) / float(r[-50:].mean())
    # per-step losses depend on the drawn noise level far more than on the weights (both sides share it), so the curves track
    # each other closely even after the weight trajectories have separated
    assert band < (0.05 if dtype_name == "fp16" else 0.10), band
    assert tail < (0.03 if dtype_name == "fp16" else 0.06), tail
    assert float(r[-50:].mean()) < 0.8 * float(r[:10].mean()) and float(m[-50:].mean()) < 0.8 * float(m[:10].mean())


def test_device_input_pipeline_bit_exact(golden_dir):
    """crop + flip + ToTensor/Normalize + ToTensorMask + fill_mask_with_random on the device (one kernel, H2D through
    pinned staging on a copy stream) against the CPU transforms of the reference's datasets, restated with torch ops in torchvision's
    operation order -- bit-exact, several batches in flight, and the result trains."""
    from joligen_amd.data_device import DeviceInputPipeline
    from pil_resize import input_pipeline_reference

    S, B, H, W = 16, 4, 24, 28
    pipe = DeviceInputPipeline(S, "cuda:0", n_buffers=2)
    g = torch.Generator().manual_seed(3)
    refs = []
    for it in range(3):
        img = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
        mask = (torch.rand(B, H, W, generator=g) < 0.3).to(torch.uint8) * torch.randint(1, 4, (B, H, W), generator=g, dtype=torch.uint8)
        off = torch.stack([torch.randint(0, H - S + 1, (B,), generator=g), torch.randint(0, W - S + 1, (B,), generator=g)], 1)
        flip = torch.rand(B, generator=g) < 0.5
        noise = torch.randn(B, 3, S, S, generator=g)
        pipe.submit(img, mask, off, flip, noise)
        # CPU restatement of the reference's transform chain: oracle/pil_resize.py::input_pipeline_reference (ToTensor, Normalize, crop,
        # hflip, fill_mask_with_random in torchvision's order of operations; torchvision itself is absent here -- the resize part of that
        # restatement is pinned against Pillow's own outputs, tests/golden/resize_pil.pt)
        refs.append(input_pipeline_reference(img, mask, off, flip, noise, S))
    for it in range(3):
        batch = pipe.get()
        torch.cuda.synchronize()
        A, Bref, Mref = refs[it]
        assert torch.equal(batch["B"].cpu(), Bref), it
        assert torch.equal(batch["B_label_mask"].cpu(), Mref) and batch["B_label_mask"].dtype == torch.int64
        assert torch.equal(batch["A"].cpu(), A), it
    # data_preprocess = "resize_and_crop" (base_dataset.py:441-443): decoded images of another size are first resized to load_size with
    # PIL's BICUBIC (masks NEAREST) -- on the device, bit-exact with Pillow (committed vectors) and with the oracle restatement
    gp = load(golden_dir, "resize_pil.pt")
    for c in gp["cases"]:
        oh, ow = c["out_hw"]
        if oh != ow:
            continue
        rp = DeviceInputPipeline(oh, "cuda:0", n_buffers=1, load_size=oh)       # crop == load: the window is the whole resized image
        rp.submit(c["img"][None], c["mask"][None], torch.zeros(1, 2, dtype=torch.int64), torch.zeros(1, dtype=torch.bool), torch.zeros(1, 3, oh, ow))
        got = rp.get()
        torch.cuda.synchronize()
        want = c["img_resized"].permute(2, 0, 1).float().div(255).sub(0.5).div(0.5)
        assert torch.equal(got["B"][0].cpu(), want), (tuple(c["img"].shape), oh)
        assert torch.equal(got["B_label_mask"][0, 0].cpu(), c["mask_resized"].long())
    L = 20
    rp = DeviceInputPipeline(S, "cuda:0", n_buffers=1, load_size=L)
    img = torch.randint(0, 256, (B, 33, 27, 3), generator=g, dtype=torch.uint8)
    mask = (torch.rand(B, 33, 27, generator=g) < 0.3).to(torch.uint8)
    off = torch.stack([torch.randint(0, L - S + 1, (B,), generator=g), torch.randint(0, L - S + 1, (B,), generator=g)], 1)
    flip = torch.rand(B, generator=g) < 0.5
    noise = torch.randn(B, 3, S, S, generator=g)
    rp.submit(img, mask, off, flip, noise)
    got = rp.get()
    torch.cuda.synchronize()
    A, Bref, Mref = input_pipeline_reference(img, mask, off, flip, noise, S, load_size=L)
    assert torch.equal(got["B"].cpu(), Bref) and torch.equal(got["A"].cpu(), A) and torch.equal(got["B_label_mask"].cpu(), Mref)
    # the batch dict is what set_input consumes
    gld = load(golden_dir, "palette_step_tiny_eff.pt")
    model = make_model(gld["cfg"], "bf16", golden_dir)
    img = torch.randint(0, 256, (2, 20, 20, 3), generator=g, dtype=torch.uint8)
    mask = torch.zeros(2, 20, 20, dtype=torch.uint8)
    mask[:, 6:14, 5:15] = 1
    pipe.submit(img, mask, torch.tensor([[2, 3], [0, 1]]))
    model.set_input(pipe.get())
    model.optimize_parameters()
    assert torch.isfinite(model.get_current_losses()["G_tot"]).all()


# ---- class-conditioned palette_model (alg_diffusion_cond_embed = "class"): oracle/make_golden_cond.py fixture ------------------------
def _cls_model(g, dtype_name, golden_dir, hp=None, **extra):
    from test_oracle_golden import cls_state
    c = g["cfg"]
    model = make_model(c, dtype_name, golden_dir, hp, alg_diffusion_cond_embed=c["cond"], alg_diffusion_cond_embed_dim=c["cond_embed_dim"],
                       alg_diffusion_dropout_prob=c["dropout_prob"], f_s_semantic_nclasses=c["nclasses"], cls_semantic_nclasses=c["nclasses"],
                       G_diff_n_timestep_test=g["sampling"]["T"], **extra)
    net = model.netG_A
    assert list(net.state_dict().keys()) == g["keys"]
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == g["shapes"]
    assert model.num_classes == g["num_classes"]
    net.load_state_dict(cls_state(g))
    return model


@pytest.mark.parametrize("dtype_name", ["fp16", "bf16"])
@pytest.mark.parametrize("tag", ["cls", "mask"])
def test_palette_class_conditioning_vs_reference_golden(golden_dir, tag, dtype_name):
    """cls: LabelEmbedder (max_norm renormalisation in place, scale_grad_by_freq) concatenated to a half-width noise-level embedding;
    mask: a per-pixel label embedding concatenated to the UNet input (38 input channels, gradient through the stem convolution into
    the table); both: the conditioning dropout of compute_palette_loss (labels AND mask of the dropped samples -> highest class), sampling:
    generator forward and DDPM restoration against the reference's outputs, 3 teacher-forced optimizer steps against the CPU oracle
    (which reproduces the reference's losses and parameter checksums of these iterations: tests/test_oracle_golden.py)."""
    import parity_util as PU
    from test_oracle_golden import cls_state, palette_conditioning_dropout

    g = load(golden_dir, f"palette_{tag}_tiny.pt")
    c, hp = g["cfg"], g["hp"]
    dtype = torch.float16 if dtype_name == "fp16" else torch.bfloat16
    model = _cls_model(g, dtype_name, golden_dir, hp)
    net = model.netG_A
    d = torch.device("cuda:0")
    k = g["table_key"]
    dev = lambda v: None if v is None else v.to(d)
    # forward
    f = g["fwd"]
    with torch.no_grad():
        noise, noise_hat, w = net(f["B"].to(d), f["A"].to(d), f["mask"].to(d), f["noise"].to(d), cls=dev(f["cls"]), t=f["t"], u=f["u"])
    assert relerr(noise_hat, f["noise_hat"]) < TOL_OUT[dtype], relerr(noise_hat, f["noise_hat"])
    row_norm = float(net.state_dict()[k][2].norm())
    assert abs(row_norm - float(f["table_row2_norm_after"])) < 1e-5, row_norm       # renormalised in place, like nn.Embedding(max_norm=1)
    # sampling with labels
    sm = g["sampling"]
    net.load_state_dict(cls_state(g))
    y, ret = net.restoration(sm["A"].to(d), y_t=sm["y_t0"].to(d), y_0=sm["B"].to(d), mask=sm["mask"].to(d), sample_num=2, cls=dev(sm["cls"]),
                             noises=sm["noises"])
    assert relerr(y, sm["y_out"]) < 2 * TOL_OUT[dtype], relerr(y, sm["y_out"])
    assert relerr(ret, sm["ret"]) < 2 * TOL_OUT[dtype]
    # steps (teacher-forced)
    net.load_state_dict(cls_state(g))
    sd0 = {kk: v.detach().float().cpu() for kk, v in net.state_dict().items()}
    tr = palette_trainer(sd0, c, hp)
    log = []
    for it, s in enumerate(g["steps"]):
        PU.force_state(net, {kk: tr.P[kk] for kk in tr.param_names}, tr.m, tr.v, tr.step, tr.ema)
        before, ref_before = PU.snapshot(net), {kk: tr.P[kk].clone() for kk in tr.param_names}
        model.rng_injection = lambda b, s=s: (s["t"], s["u"], s["noise"])
        model.drop_injection = lambda b, s=s: s["drop_u"]
        data = {"A": s["A"], "B": s["B"], "B_label_mask": s["mask"], "A_img_paths": ["x"]}
        if s["cls"] is not None:
            data["B_label_cls"] = s["cls"]
        model.set_input(data)
        model.optimize_parameters()
        loss = float(model.get_current_losses()["G_tot"])
        cls, mask = palette_conditioning_dropout(s["drop_u"], c["dropout_prob"], g["num_classes"], s["cls"], s["mask"])
        loss_ref = float(tr.optimize_parameters(s["B"], s["A"], mask, s["noise"], s["t"], s["u"], cls=cls))
        assert abs(loss_ref - float(s["loss"])) < 2e-4 * abs(float(s["loss"])) + 1e-6
        assert abs(loss - loss_ref) < TOL_LOSS_FWD[dtype] * abs(loss_ref), (it, loss, loss_ref)
        after = PU.snapshot(net)
        PU.check_update(f"{tag} {dtype_name} it{it}", before, after, ref_before, {kk: tr.P[kk] for kk in tr.param_names}, COS_UPDATE[dtype], log=log)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/update_agreement_palette_{tag}_{dtype_name}.txt", "w") as fh:
        fh.write("\n".join(log))


def test_palette_class_embedding_gradient_vs_oracle(golden_dir):
    """first-step gradient of the label-embedding table (rows scaled by 1 / their frequency in the batch, untouched rows zero) and of
    the half-width cond_embed MLP against the CPU oracle"""
    g = load(golden_dir, "palette_cls_tiny.pt")
    c = g["cfg"]
    model = _cls_model(g, "fp16", golden_dir, g["hp"], train_G_ema=False)
    net = model.netG_A
    s = g["steps"][0]
    cls = torch.tensor([3, 3])              # the same class twice: scale_grad_by_freq halves the row's gradient
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tr = O.OraclePaletteTrainer(sd, cfg_of(c), ema_beta=None)
    loss_ref, grads_ref, _ = tr.loss_and_grads(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"], cls=cls)
    model.rng_injection = lambda b: (s["t"], s["u"], s["noise"])
    model.drop_injection = lambda b: torch.ones(b)          # nothing dropped
    model.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"], "B_label_cls": cls})
    model.compute_palette_loss()
    model.loss_G_tot.backward()
    torch.cuda.synchronize()
    assert abs(float(model.loss_G_tot) - float(loss_ref)) < 5e-3 * float(loss_ref)
    k = "denoise_fn.netl_embedder_class.embedding_table.weight"
    mine = dict(net.named_parameters())[k].grad.detach().float().cpu() / model.loss_scale
    ref = grads_ref[k]
    assert float(ref[3].norm()) > 0 and float(ref[[0, 1, 2, 4, 5]].abs().max()) == 0.0
    assert float(mine[[0, 1, 2, 4, 5]].abs().max()) == 0.0
    assert relerr(mine[3], ref[3]) < TOL_GRAD[torch.float16], relerr(mine[3], ref[3])
    for kk in ("cond_embed.0.weight", "cond_embed.2.weight"):
        e = relerr(dict(net.named_parameters())[kk].grad / model.loss_scale, grads_ref[kk])
        assert e < TOL_GRAD[torch.float16], (kk, e)


def test_palette_mask_embedding_gradient_vs_oracle(golden_dir):
    """first-step gradient of the per-pixel mask-embedding table (through the input gradient of the 38-channel stem convolution; rows
    scaled by 1 / their pixel count in the batch, unused classes zero) and of the stem weights against the CPU oracle"""
    g = load(golden_dir, "palette_mask_tiny.pt")
    c = g["cfg"]
    model = _cls_model(g, "fp16", golden_dir, g["hp"], train_G_ema=False)
    net = model.netG_A
    s = g["steps"][0]
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    tr = O.OraclePaletteTrainer(sd, cfg_of(c), ema_beta=None)
    loss_ref, grads_ref, _ = tr.loss_and_grads(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"])
    model.rng_injection = lambda b: (s["t"], s["u"], s["noise"])
    model.drop_injection = lambda b: torch.ones(b)
    model.set_input({"A": s["A"], "B": s["B"], "B_label_mask": s["mask"]})
    model.compute_palette_loss()
    model.loss_G_tot.backward()
    torch.cuda.synchronize()
    assert abs(float(model.loss_G_tot) - float(loss_ref)) < 5e-3 * float(loss_ref)
    k = g["table_key"]
    mine = dict(net.named_parameters())[k].grad.detach().float().cpu() / model.loss_scale
    ref = grads_ref[k]
    used = sorted(set(int(v) for v in s["mask"].unique()))
    unused = [i for i in range(ref.shape[0]) if i not in used]
    assert float(ref[used].norm()) > 0 and float(ref[unused].abs().max()) == 0.0 and float(mine[unused].abs().max()) == 0.0
    assert relerr(mine[used], ref[used]) < 2 * TOL_GRAD[torch.float16], relerr(mine[used], ref[used])
    kk = "denoise_fn.model.input_blocks.0.0.weight"
    e = relerr(dict(net.named_parameters())[kk].grad / model.loss_scale, grads_ref[kk])
    assert e < TOL_GRAD[torch.float16], e


@pytest.mark.parametrize("dtype_name", ["bf16", "fp16"])
def test_full_size_batch32_step_equals_single_sample_steps(golden_dir, dtype_name):
    """BASELINE configs[1] at its FULL size (ngf 64, mults [1,2,4,8], 256 x 256, batch 32 -- the bench workload, where the dispatch picks
    the 256-wide 8-wave tiles, the persistent 64-channel kernel, the split-K 308 / 171 weight gradients): a size-independent property
    ties it to the batch-1 case that IS compared with the CPU oracle (test_first_step_gradients_vs_oracle_baseline_shapes).  GroupNorm
    and the loss are per sample, so loss(batch) = mean_i loss(sample i) and grad(batch) = 1/32 sum_i grad(sample i); the 32 single-sample
    passes run through the small-grid kernel configurations."""
    c = dict(C2, B=32)
    model = make_model(c, dtype_name, golden_dir, train_G_ema=False)
    net = model.netG_A
    B, S = 32, 256
    g = torch.Generator().manual_seed(21)
    Bimg = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    mask = torch.zeros(B, 1, S, S, dtype=torch.int64)
    for i in range(B):
        h0, w0 = 8 * (i % 11), 16 * (i % 7)
        mask[i, :, h0:h0 + 64 + 4 * i, w0:w0 + 96] = 1
    A = Bimg * (1 - mask) + torch.randn(B, 3, S, S, generator=g) * mask
    t, u, noise = O.draw_step_randomness(torch.Generator().manual_seed(22), Bimg, 2000)

    def run(sl):
        net.arena.g.zero_()
        model.rng_injection = lambda b: (t[sl], u[sl], noise[sl])
        model.set_input({"A": A[sl], "B": Bimg[sl], "B_label_mask": mask[sl]})
        model.compute_palette_loss()
        model.loss_G_tot.backward()
        torch.cuda.synchronize()
        return float(model.loss_G_tot), net.arena.g.clone() / model.loss_scale

    loss32, g32 = run(slice(0, B))
    assert torch.isfinite(g32).all()
    acc, losses = torch.zeros_like(g32), []
    for i in range(B):
        li, gi = run(slice(i, i + 1))
        losses.append(li)
        acc += gi
    acc /= B
    fp16 = dtype_name == "fp16"
    assert abs(loss32 - sum(losses) / B) < (2e-3 if fp16 else 8e-3) * loss32, (loss32, sum(losses) / B)
    e = relerr(g32, acc)
    assert e < (1e-2 if fp16 else 5e-2), e
    # per tensor: every convolution weight of the network (all tile configurations of the dispatch appear among them)
    bad = []
    for name, p in net.named_parameters():
        if p.dim() == 4:
            off, n = net.arena.slices[name]
            a, b_ = g32[off:off + n], acc[off:off + n]
            if float(b_.norm()) > 0 and relerr(a, b_) > (3e-2 if fp16 else 0.15):
                bad.append((name, relerr(a, b_)))
    assert not bad, bad[:8]


@pytest.mark.parametrize("efficient", [True, False])
def test_gn_backward_with_fused_coefficients(golden_dir, efficient, monkeypatch):
    """jg_gn_bwd_apply_fc (coefficient step inside the apply pass; plain, FiLM, pooled-gradient and addend forms all occur in this
    network) against the three-launch form: same input gradient, embedding gradient and gradient arena"""
    from joligen_amd import ops
    from joligen_amd.modules import unet_exec

    c = dict(ngf=64, mults=[1, 2], res_blocks=[1, 1], attn_res=[2], efficient=efficient, S=64, B=2)
    dtype = torch.float16
    net, _ = build_net(c, dtype, golden_dir)
    unet = net.denoise_fn.model
    net.arena.ensure_fresh()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    x0 = ops.to_nhwc(torch.randn(c["B"], 6, c["S"], c["S"], generator=g).to(d), dtype, 8)
    emb0 = torch.randn(c["B"], unet.cond_embed_dim, generator=g).to(d)
    R = ops.to_nhwc(torch.randn(c["B"], 3, c["S"], c["S"], generator=g).to(d), dtype, 8)
    res = {}
    for fc in (False, True):
        monkeypatch.setattr(unet_exec, "FUSE_GN_COEF", fc)
        net.arena.g.zero_()
        x = x0.clone().requires_grad_(True)
        emb = emb0.clone().requires_grad_(True)
        unet(x, emb).backward(R)
        torch.cuda.synchronize()
        res[fc] = (x.grad.float(), emb.grad.clone(), net.arena.g.clone())
    for i, name in enumerate(("dx", "demb", "gradient arena")):
        e = relerr(res[True][i], res[False][i])
        assert e < 5e-3, (name, e)        # same arithmetic, other summation orders (fp32 atomics) in front of fp16 stores


@pytest.mark.parametrize("graph", ["fused", "modules", "torch_ops"])
@pytest.mark.parametrize("dtype_name", ["bf16", "fp16"])
def test_palette_step_is_bit_reproducible_in_deterministic_mode(golden_dir, dtype_name, graph):
    """JG_DETERMINISTIC=1 (palette family): GroupNorm statistics and backward reductions from ordered single-workgroup passes
    (no fused epilogue statistics), GroupNorm parameter gradients summed over the images in order, weight gradients without a split over the
    pixels, one-workgroup loss reductions, no K split in the fp32 GEMM.  THREE optimize_parameters() calls from the same state, twice: loss,
    every gradient, every parameter, Adam moment and EMA weight after each call are torch.equal -- on the fused schedule (the benchmarked
    graph, a torch.ops node), on the module-by-module graph and through the `torch.ops` boundary.  And the mode computes the same step: against
    the default (atomic) mode the loss agrees to 1.5e-3 (fp16) / 8e-3 (bf16) and the update keeps its direction and length."""
    import contextlib

    import parity_util as PU
    from joligen_amd import _lib, ops

    c = dict(ngf=32, mults=[1, 2, 4], res_blocks=[1, 1, 1], attn_res=[4], efficient=True, S=64, B=3)
    g = torch.Generator().manual_seed(5)
    S, B = c["S"], c["B"]
    Bimg = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    mask = torch.zeros(B, 1, S, S, dtype=torch.int64)
    mask[:, :, 6:40, 10:50] = 1
    A = Bimg * (1 - mask) + torch.randn(B, 3, S, S, generator=g) * mask
    draws = [O.draw_step_randomness(torch.Generator().manual_seed(9 + i), Bimg, 2000) for i in range(3)]

    def run(det, graph=graph):
        prev = _lib.set_tuning("JG_DETERMINISTIC", det)
        try:
            model = make_model(c, dtype_name, golden_dir, train_G_ema=True)
            model.netG_A.denoise_fn.model.jg_fused = graph == "fused"
            out = []
            for i in range(3):
                model.rng_injection = lambda b, i=i: draws[i]
                model.set_input({"A": A, "B": Bimg, "B_label_mask": mask})
                with (ops.torch_ops_boundary() if graph == "torch_ops" else contextlib.nullcontext()):
                    model.optimize_parameters()
                torch.cuda.synchronize()
                ar = model.netG_A.arena
                out.append(dict(loss=model.loss_G_tot.detach().clone(), p=ar.p.clone(), m=ar.m.clone(), v=ar.v.clone(),
                                ema=ar.ema.clone()))
            return out
        finally:
            _lib.set_tuning("JG_DETERMINISTIC", prev)

    a, b = run(1), run(1)
    for i, (x, y) in enumerate(zip(a, b)):
        for k in x:
            assert torch.equal(x[k], y[k]), (graph, "call", i, k, float((x[k].double() - y[k].double()).abs().max()))
    if graph == "torch_ops":
        # ops-vs-ctypes without a run-to-run floor in the way (what test_palette_step_through_torch_ops bounds by 3 x a measured floor): the
        # FORWARD is the same kernels on the same bits -- the first loss is torch.equal; the backward is not the same arithmetic (the op
        # graph sums a tensor's two gradients with autograd's 16-bit add and accumulates fresh dw tensors into .grad, the ctypes nodes fold
        # the addends into the GroupNorm-backward pass in fp32 and accumulate in the arena), so the update agrees to rounding, not to the bit
        m_ = run(1, "modules")
        assert torch.equal(a[0]["loss"], m_[0]["loss"])
        e = float((a[0]["m"].double() - m_[0]["m"].double()).norm() / m_[0]["m"].double().norm())
        assert e < 1e-6, e          # measured 1.2e-8 / 1.3e-8: fp32 rounding of the differently grouped sums, no 16-bit event flips
        with open(f"gpurun_out/deterministic_ops_vs_ctypes_{dtype_name}.txt", "w") as f:
            f.write(f"first loss equal: True; first moments (all gradients as one vector) torch.ops vs ctypes nodes, deterministic mode: {e:.3e}\n")
    d = run(0)
    for i in range(3):
        # (the ordered statistics pass reads the 16-bit output of the convolution, the fused epilogue its fp32 accumulators)
        assert abs(float(a[i]["loss"]) - float(d[i]["loss"])) <= (1.5e-3 if dtype_name == "fp16" else 8e-3) * abs(float(d[i]["loss"])) + 1e-6
    # first moments after the first call = (1 - beta1) x gradient: same direction and length as in the default mode
    ga, gd = a[0]["m"].double(), d[0]["m"].double()
    cos = float((ga * gd).sum() / (ga.norm() * gd.norm()))
    # (a LOOSE sanity bound, not a parity claim: the two modes round the GroupNorm statistics at different points, and in bf16 the gradient of a
    #  random-weight UNet moves by 1 - 2 % in length for that; measured 1.022 at worst)
    lim = 0.02 if dtype_name == "fp16" else 0.04
    assert cos > (0.999 if dtype_name == "fp16" else 0.99) and 1 - lim < float(ga.norm() / gd.norm()) < 1 + lim, (cos, float(ga.norm() / gd.norm()))


@pytest.mark.parametrize("efficient", [True, False])
def test_fused_unet_node_is_a_torch_op(golden_dir, efficient, monkeypatch):
    """The benchmarked palette graph goes through `torch.ops` (VERDICT r5 missing #3): UNet.forward on the fused schedule is
    `torch.ops.jg355.unet_fused` (+ `unet_fused_bwd` through register_autograd), by default.  Held against the autograd.Function of rounds 1-5 on
    the same executor (JG_FUSED_TORCH_OPS=0): output, input gradient, embedding gradient and the gradient arena to the summation-order
    floor; schema + fake-tensor checks of torch.library.opcheck on both ops; an inference call keeps no tape and a dropped graph frees its tape."""
    import gc

    from joligen_amd import ops
    from joligen_amd.modules import unet_exec

    c = dict(ngf=64, mults=[1, 2], res_blocks=[1, 1], attn_res=[2], efficient=efficient, S=64, B=2)
    dtype = torch.float16
    net, _ = build_net(c, dtype, golden_dir)
    unet = net.denoise_fn.model
    net.arena.ensure_fresh()
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    x0 = ops.to_nhwc(torch.randn(c["B"], 6, c["S"], c["S"], generator=g).to(d), dtype, 8)
    emb0 = torch.randn(c["B"], unet.cond_embed_dim, generator=g).to(d)
    R = ops.to_nhwc(torch.randn(c["B"], 3, c["S"], c["S"], generator=g).to(d), dtype, 8)
    res = {}
    for as_op in (False, True, False, False, False):       # the floor below is the largest of three repeats (a single repeat is a noisy yardstick)
        monkeypatch.setattr(unet_exec, "FUSED_TORCH_OPS", as_op)
        net.arena.g.zero_()
        x = x0.clone().requires_grad_(True)
        emb = emb0.clone().requires_grad_(True)
        y = unet(x, emb)
        assert ("Unet_fused" in type(y.grad_fn).__name__ or "unet_fused" in type(y.grad_fn).__name__.lower()) == as_op, type(y.grad_fn).__name__
        y.backward(R)
        torch.cuda.synchronize()
        res.setdefault(as_op, []).append((y.detach().float(), x.grad.float(), emb.grad.clone(), net.arena.g.clone()))
    assert not unet_exec._TAPES, "the backward consumed its tape"
    for i, name in enumerate(("y", "dx", "demb", "gradient arena")):
        floor = max(relerr(r[i], res[False][0][i]) for r in res[False][1:])
        e = relerr(res[True][0][i], res[False][0][i])
        assert e <= 3 * floor + 1e-6, (name, e, floor)
    assert float(res[True][0][3].norm()) > 0
    # inference: no tape; a graph dropped without a backward: its tape goes with it
    monkeypatch.setattr(unet_exec, "FUSED_TORCH_OPS", True)
    with torch.no_grad():
        unet(x0, emb0)
    assert not unet_exec._TAPES
    y = unet(x0.clone().requires_grad_(True), emb0)
    assert len(unet_exec._TAPES) == 1
    del y
    gc.collect()
    assert not unet_exec._TAPES
    exe = unet._jg_executor
    emb_all = unet._emb_all(emb0.float().contiguous()).all.detach()
    torch.library.opcheck(torch.ops.jg355.unet_fused, (x0, emb_all, net.arena.w16, net.arena.w16T, exe.handle, False), test_utils=("test_schema", "test_faketensor"))
    out, tape = torch.ops.jg355.unet_fused(x0, emb_all, net.arena.w16, net.arena.w16T, exe.handle, True)
    with pytest.raises(ValueError):          # the mutated argument is the network's own gradient arena, nothing else
        torch.ops.jg355.unet_fused_bwd(R, tape, torch.zeros_like(net.arena.g), exe.handle, True)
    net.arena.g.zero_()
    dx, demb = torch.ops.jg355.unet_fused_bwd(R, tape, net.arena.g, exe.handle, True)
    torch.cuda.synchronize()
    assert dx.shape == x0.shape and demb.shape == emb_all.shape and float(net.arena.g.norm()) > 0 and not unet_exec._TAPES


@pytest.mark.parametrize("dtype_name", ["bf16", "fp16"])
def test_palette_step_through_torch_ops(golden_dir, dtype_name):
    """The op boundary of north_star / SURVEY.md 8(b3): ONE palette training step with every op a `torch.ops.jg355.*` call
    (`ops.torch_ops_boundary()`: conv2d_nt on the fp32 master weights, group_norm_act, attention_core, resample2, linear_act,
    gamma_embedding, ddpm_prepare, ddpm_mse_loss; concatenation = torch.cat; autograd assembles the backward from the registered
    formulas) against the same step on the module-by-module graph of ctypes autograd nodes: same kernels behind both, so loss and
    every parameter gradient agree to the run-to-run floor of the ctypes graph itself (measured here), and after
    `optimize_parameters()` through the boundary the update has the same direction and length.  (The default step is the fused schedule: one autograd node)"""
    import parity_util as PU
    from joligen_amd import ops

    c = dict(ngf=32, mults=[1, 2, 4], res_blocks=[1, 1, 1], attn_res=[4], efficient=True, S=32, B=2)
    g = torch.Generator().manual_seed(5)
    Bimg = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    mask = torch.zeros(2, 1, 32, 32, dtype=torch.int64)
    mask[:, :, 6:20, 10:26] = 1
    A = Bimg * (1 - mask) + torch.randn(2, 3, 32, 32, generator=g) * mask
    t, u, noise = O.draw_step_randomness(torch.Generator().manual_seed(9), Bimg, 2000)

    def run(boundary, full_step=False):
        model = make_model(c, dtype_name, golden_dir, train_G_ema=False)
        model.netG_A.denoise_fn.model.jg_fused = False           # module-by-module graph on both sides
        model.rng_injection = lambda b: (t, u, noise)
        model.set_input({"A": A, "B": Bimg, "B_label_mask": mask})
        before = PU.snapshot(model.netG_A)
        ctx = ops.torch_ops_boundary() if boundary else contextlib.nullcontext()
        with ctx:
            if full_step:
                model.optimize_parameters()
                torch.cuda.synchronize()
                return float(model.loss_G_tot.detach()), before, PU.snapshot(model.netG_A)
            model.compute_palette_loss()
            model.loss_G_tot.backward()
        torch.cuda.synchronize()
        return float(model.loss_G_tot.detach()), {k: p.grad.detach().float().cpu().clone() for k, p in model.netG_A.named_parameters()}

    import contextlib

    # yardstick measured on the spot: the SAME ctypes graph run twice.  GroupNorm statistics and split-K weight gradients are summed with
    # fp32 atomics, so two runs differ in the last bit of a few sums, a handful of 16-bit activations round the other way, and the step
    # is reproducible only to that floor -- the torch.ops form must sit on it (it launches the same kernels on the same bits).
    # (round 6: plain 1x1 layers with >= 256 channels go to the LDS-tiled GEMM by default while the fused schedule's GroupNorm forms of the same
    #  layers exist only in the streaming kernel; JG_CONV1X1 = 2 keeps every 1x1 layer of BOTH graphs on the streaming kernel, so that this stays
    #  a comparison of the same kernels)
    from joligen_amd import _lib as _jl

    prev_1x1 = _jl.set_tuning("JG_CONV1X1", 2)
    try:
        loss_c, grads_c = run(False)
        again = [run(False) for _ in range(3)]
        loss_o, grads_o = run(True)
    finally:
        _jl.set_tuning("JG_CONV1X1", prev_1x1)
    # the floor is the LARGEST of three repeats: one repeat of a scalar (the loss) lands next to the first run by luck often enough to fail a
    # full-suite run now and then (seen once in round 6: floor 0, 2e-4 of slack, bf16 noise 6e-4); the fixed slack is the dtype's
    # own run-to-run scale
    loss_c2, grads_c2 = again[0]
    keys = [k for k in grads_c if float(grads_c[k].norm()) > 0]
    slack_loss, slack_whole = (2e-3, 2e-2) if dtype_name == "bf16" else (3e-4, 3e-3)
    floor_loss = max(abs(l2 - loss_c) / abs(loss_c) for l2, _ in again)
    floor_grad = max(relerr(g2[k], grads_c[k]) for _, g2 in again for k in keys)
    assert abs(loss_o - loss_c) <= (3 * floor_loss + slack_loss) * abs(loss_c), (loss_o, loss_c, [l2 for l2, _ in again])
    worst = max((relerr(grads_o[k], grads_c[k]), k) for k in keys)
    assert worst[0] <= 3 * floor_grad + 2e-3, (worst, floor_grad)       # (the floor's worst tensor is a conv bias in front of a GroupNorm: pure noise)
    cat = lambda gr: torch.cat([gr[k].flatten() for k in keys])
    whole, floor_whole = relerr(cat(grads_o), cat(grads_c)), max(relerr(cat(g2), cat(grads_c)) for _, g2 in again)
    assert whole <= 3 * floor_whole + slack_whole, (whole, floor_whole)   # all parameter gradients as one vector
    zero = [k for k in grads_c if (float(grads_c[k].norm()) == 0) != (float(grads_o[k].norm()) == 0)]
    assert not zero, zero
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/torch_ops_step_{dtype_name}.txt", "w") as f:
        f.write(f"loss ctypes {loss_c!r} ctypes again {loss_c2!r} torch.ops {loss_o!r}\nworst gradient tensor torch.ops vs ctypes {worst}\n"
                f"run-to-run floor of the ctypes graph: loss {floor_loss:.3e} worst tensor {floor_grad:.3e}\n"
                f"all gradients as one vector: torch.ops vs ctypes {whole:.3e}, ctypes run-to-run {floor_whole:.3e}\n")
    # the whole optimize_parameters() through the boundary: same update as through the ctypes nodes
    _, b_c, a_c = run(False, full_step=True)
    _, b_o, a_o = run(True, full_step=True)
    cos, ratio, _ = PU.update_agreement(b_o, a_o, b_c, a_c)
    assert cos > 0.9 and 0.95 < ratio < 1.05, (cos, ratio)

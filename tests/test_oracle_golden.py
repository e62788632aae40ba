"""CPU: the oracle restatement (oracle/jg_oracle.py) against fixtures produced by the
unmodified reference (oracle/make_golden.py).  This is what pins the oracle."""
import os

import pytest
import torch

import jg_oracle as O

CFGS = ["tiny_eff", "tiny_noeff", "tiny_attn"]


def cfg_of(c):
    # mask conditioning (alg_diffusion_cond_embed containing "mask") adds cond_embed_dim input channels (diffusion_networks.py:112-113)
    extra = c.get("cond_embed_dim", 0) if "mask" in c.get("cond", "") else 0
    return O.UNetCfg(in_channel=6 + extra, inner_channel=c["ngf"], out_channel=3, res_blocks=c["res_blocks"],
                     attn_res=c["attn_res"], channel_mults=c["mults"], efficient=c["efficient"])


def load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def synth_for(golden_dir, name):
    g = load(golden_dir, f"palette_step_{name}.pt")
    sched = load(golden_dir, "schedule.pt")
    ref_sd = {}
    for k in g["keys"]:
        leaf = k.split(".")[-1]
        ref_sd[k] = sched[leaf] if O._is_buffer(k) else torch.empty(g["shapes"][k])
    return O.synth_state_dict(ref_sd, seed=0), g


def test_schedule_buffers(golden_dir):
    sched = load(golden_dir, "schedule.pt")
    mine = {}
    mine.update(O.noise_schedule_buffers("train", 2000))
    mine.update(O.noise_schedule_buffers("test", 1000))
    assert set(mine) == set(sched)
    for k in sched:
        assert torch.equal(mine[k], sched[k]), k  # float64 numpy -> fp32: bit exact


@pytest.mark.parametrize("name", CFGS)
def test_unet_forward_backward(golden_dir, name):
    sd, _ = synth_for(golden_dir, name)
    g = load(golden_dir, f"unet_{name}.pt")
    cfg = cfg_of(g["cfg"])
    pre = "denoise_fn.model."
    P = {k[len(pre):]: v.clone().requires_grad_(not O._is_buffer(k)) for k, v in sd.items() if k.startswith(pre)}
    x = g["x"].clone().requires_grad_(True)
    emb = g["emb"].clone().requires_grad_(True)
    out = O.unet_forward(P, x, emb, cfg)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    (out * g["R"]).sum().backward()
    torch.testing.assert_close(x.grad, g["dx"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(emb.grad, g["demb"], rtol=1e-4, atol=1e-3)
    for k, ref in g["grad_checks"].items():
        v = P[k].grad
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        torch.testing.assert_close(mine, ref, rtol=2e-4, atol=2e-4 * float(ref[0]) + 1e-6, msg=k)


def test_unet_attention_heads_of_16_channels(golden_dir):
    """`G_unet_mha_num_head_channels = 16` (what the reference's own run tests select, tests/test_run_diffusion.py:24): 64 channels = 4 heads
    of 16 in every AttentionBlock.  UNet forward + backward of the unmodified reference (oracle/make_golden_heads16.py) vs the restatement."""
    g = load(golden_dir, "unet_heads16.pt")
    sched = load(golden_dir, "schedule.pt")
    sd = O.synth_state_dict({k: (sched[k.split(".")[-1]] if O._is_buffer(k) else torch.empty(g["shapes"][k])) for k in g["keys"]}, seed=0)
    c = g["cfg"]
    cfg = O.UNetCfg(in_channel=6, inner_channel=c["ngf"], out_channel=3, res_blocks=c["res_blocks"], attn_res=c["attn_res"], channel_mults=c["mults"],
                    efficient=c["efficient"], num_head_channels=c["num_head_channels"])
    pre = "denoise_fn.model."
    P = {k[len(pre):]: v.clone().requires_grad_(not O._is_buffer(k)) for k, v in sd.items() if k.startswith(pre)}
    x, emb = g["x"].clone().requires_grad_(True), g["emb"].clone().requires_grad_(True)
    out = O.unet_forward(P, x, emb, cfg)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    (out * g["R"]).sum().backward()
    torch.testing.assert_close(x.grad, g["dx"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(emb.grad, g["demb"], rtol=1e-4, atol=1e-3)
    for k, ref in g["grad_checks"].items():
        v = P[k].grad
        torch.testing.assert_close(torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()]), ref, rtol=2e-4, atol=2e-4 * float(ref[0]) + 1e-6, msg=k)
    # and it is not the 32-channel split in disguise
    out32 = O.unet_forward({k: v.detach() for k, v in P.items()}, g["x"], g["emb"], O.UNetCfg(in_channel=6, inner_channel=c["ngf"], out_channel=3,
                           res_blocks=c["res_blocks"], attn_res=c["attn_res"], channel_mults=c["mults"], efficient=c["efficient"], num_head_channels=32))
    assert float((out32 - g["out"]).norm() / g["out"].norm()) > 1e-3


@pytest.mark.parametrize("name", CFGS)
def test_diffusion_generator_forward(golden_dir, name):
    sd, _ = synth_for(golden_dir, name)
    g = load(golden_dir, f"diffgen_{name}.pt")
    cfg = cfg_of(g["cfg"])
    with torch.no_grad():
        noise, noise_hat, w, _ = O.diffusion_generator_forward(sd, g["B"], g["A"], g["mask"], g["noise"], g["t"], g["u"], cfg)
    torch.testing.assert_close(noise_hat, g["noise_hat"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(w, g["min_snr_w"], rtol=1e-6, atol=0)
    # the generator draw order is the reference's
    gen = torch.Generator().manual_seed(77)
    t, u, n = O.draw_step_randomness(gen, g["B"], 2000)
    assert torch.equal(t, g["t"]) and torch.equal(u, g["u"]) and torch.equal(n, g["noise"])


@pytest.mark.parametrize("name", CFGS)
def test_palette_three_steps(golden_dir, name):
    sd, g = synth_for(golden_dir, name)
    cfg = cfg_of(g["cfg"])
    hp = g["hp"]
    tr = O.OraclePaletteTrainer(sd, cfg, lr=hp["lr"], beta1=hp["beta1"], beta2=hp["beta2"], eps=hp["eps"],
                                weight_decay=hp["weight_decay"], ema_beta=hp["ema_beta"],
                                lambda_G=hp["lambda_G"], optim=hp["optim"])
    for it, s in enumerate(g["steps"]):
        loss = tr.optimize_parameters(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"])
        torch.testing.assert_close(loss, s["loss"], rtol=2e-4, atol=1e-6)
        if "param_checks" in s:
            for k, ref in s["param_checks"].items():
                v = tr.P[k]
                mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
                torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=k)
            for k, ref in s["ema_checks"].items():
                v = tr.ema[k]
                mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
                torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=k)
    for k, ref in g["param_sample"].items():
        torch.testing.assert_close(tr.P[k].flatten()[:8], ref, rtol=1e-3, atol=2e-5, msg=k)


def test_palette_pix2pix_three_steps(golden_dir):
    """alg_diffusion_task = "pix2pix" (models/palette_model.py:360-363; examples/example_ddpm_SEN2VEN.json): conditioning on the paired
    image A, no mask -- no ground-truth blend of the noisy image, loss over every pixel.  Three optimize_parameters() of the unmodified
    reference (oracle/make_golden_pix2pix.py) against the CPU restatement."""
    g = load(golden_dir, "palette_step_pix2pix_tiny.pt")
    sched = load(golden_dir, "schedule.pt")
    sd = O.synth_state_dict({k: (sched[k.split(".")[-1]] if O._is_buffer(k) else torch.empty(g["shapes"][k])) for k in g["keys"]}, seed=0)
    hp = g["hp"]
    tr = O.OraclePaletteTrainer(sd, cfg_of(g["cfg"]), lr=hp["lr"], beta1=hp["beta1"], beta2=hp["beta2"], eps=hp["eps"], weight_decay=hp["weight_decay"],
                                ema_beta=hp["ema_beta"], lambda_G=hp["lambda_G"], optim=hp["optim"])
    for s in g["steps"]:
        loss = tr.optimize_parameters(s["B"], s["A"], None, s["noise"], s["t"], s["u"])
        torch.testing.assert_close(loss, s["loss"], rtol=2e-4, atol=1e-6)
        for which, store in (("param_checks", tr.P), ("ema_checks", tr.ema)):
            for k, ref in s.get(which, {}).items():
                v = store[k]
                mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
                torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=k)


def test_palette_minsnr_three_steps(golden_dir):
    """`alg_palette_minsnr = True` (what tests/test_run_diffusion.py of the reference selects): min(SNR, 5) / SNR per sample on both
    operands of the loss.  Three optimize_parameters() of the unmodified reference (oracle/make_golden_minsnr.py) against the restatement."""
    g = load(golden_dir, "palette_step_minsnr_tiny.pt")
    sched = load(golden_dir, "schedule.pt")
    sd = O.synth_state_dict({k: (sched[k.split(".")[-1]] if O._is_buffer(k) else torch.empty(g["shapes"][k])) for k in g["keys"]}, seed=0)
    hp = g["hp"]
    tr = O.OraclePaletteTrainer(sd, cfg_of(g["cfg"]), lr=hp["lr"], beta1=hp["beta1"], beta2=hp["beta2"], eps=hp["eps"], weight_decay=hp["weight_decay"],
                                ema_beta=hp["ema_beta"], lambda_G=hp["lambda_G"], optim=hp["optim"], minsnr=True)
    plain = O.OraclePaletteTrainer(sd, cfg_of(g["cfg"]), lambda_G=hp["lambda_G"])
    for it, s in enumerate(g["steps"]):
        if it == 0:      # the weight matters: the unweighted loss of the same step is another number
            assert abs(float(plain.loss_and_grads(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"])[0]) - float(s["loss"])) > 1e-3 * float(s["loss"])
        loss = tr.optimize_parameters(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"])
        torch.testing.assert_close(loss, s["loss"], rtol=2e-4, atol=1e-6)
        for which, store in (("param_checks", tr.P), ("ema_checks", tr.ema)):
            for k, ref in s.get(which, {}).items():
                v = store[k]
                mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
                torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=k)


def test_palette_gradient_accumulation(golden_dir):
    """`train_iter_size = 2` (models/base_model.py:1250-1282,1302-1377; the shipped DDPM example trains with 16): four calls of the
    unmodified reference's optimize_parameters() = two optimizer steps (oracle/make_golden_accum.py).  Pinned per call: the raw loss, the
    parameters (unchanged on a non-boundary call), the EMA copy (updated on EVERY call), and the `G_tot_avg` the loss log reports."""
    g = load(golden_dir, "palette_step_accum_tiny.pt")
    sched = load(golden_dir, "schedule.pt")
    sd = O.synth_state_dict({k: (sched[k.split(".")[-1]] if O._is_buffer(k) else torch.empty(g["shapes"][k])) for k in g["keys"]}, seed=0)
    hp = g["hp"]
    tr = O.OraclePaletteTrainer(sd, cfg_of(g["cfg"]), lr=hp["lr"], beta1=hp["beta1"], beta2=hp["beta2"], eps=hp["eps"], weight_decay=hp["weight_decay"],
                                ema_beta=hp["ema_beta"], lambda_G=hp["lambda_G"], optim=hp["optim"])
    start = {k: tr.P[k].clone() for k in tr.param_names}
    for it, s in enumerate(g["steps"]):
        loss, reported = tr.iteration(s["B"], s["A"], s["mask"], s["noise"], s["t"], s["u"], iter_size=g["iter_size"])
        torch.testing.assert_close(loss, s["loss_raw"], rtol=2e-4, atol=1e-6)
        if it == 0:      # first call of a window: nothing stepped yet
            assert all(torch.equal(tr.P[k], start[k]) for k in tr.param_names)
        assert (reported is not None) == ("losses_reported" in s)
        if reported is not None:
            torch.testing.assert_close(reported, s["losses_reported"]["G_tot_avg"], rtol=2e-4, atol=1e-6)
        for which, store in (("param_checks", tr.P), ("ema_checks", tr.ema)):
            for k, ref in s[which].items():
                v = store[k]
                mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
                torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=f"{which} it{it} {k}")


# ---- consistency model (cm_model): oracle/make_golden_cm.py fixtures --------------------------------------
CM_CFGS = ["tiny_eff", "tiny_attn"]


def cm_cfg_of(c):
    return O.UNetCfg(in_channel=3, inner_channel=c["ngf"], out_channel=3, res_blocks=c["res_blocks"],
                     attn_res=c["attn_res"], channel_mults=c["mults"], efficient=c["efficient"], cond_embed_dim=256)


def cm_synth_for(golden_dir, name):
    g = load(golden_dir, f"cm_step_{name}.pt")
    ref_sd = {k: torch.empty(g["shapes"][k]) for k in g["keys"]}
    return O.synth_state_dict(ref_sd, seed=0), g


@pytest.mark.parametrize("name", CM_CFGS)
def test_cm_generator_forward(golden_dir, name):
    sd, _ = cm_synth_for(golden_dir, name)
    g = load(golden_dir, f"cm_gen_{name}.pt")
    cfg = cm_cfg_of(g["cfg"])
    with torch.no_grad():
        out = O.cm_generator_forward(sd, g["B"], g["mask"], g["noise"], g["timesteps"], 0, g["total_t"], cfg)
    assert out[2] == g["num_timesteps"]
    assert torch.equal(out[3], g["sigmas"])
    torch.testing.assert_close(out[4], g["loss_weights"], rtol=1e-6, atol=0)
    torch.testing.assert_close(out[5], g["next_noisy_x"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out[6], g["current_noisy_x"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out[0], g["next_x"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out[1], g["current_x"], rtol=1e-4, atol=1e-5)
    noise, ts = O.cm_draw_step_randomness(torch.Generator().manual_seed(55), g["B"], g["sigmas"])
    assert torch.equal(noise, g["noise"]) and torch.equal(ts, g["timesteps"])


@pytest.mark.parametrize("name", CM_CFGS)
def test_cm_three_steps(golden_dir, name):
    sd, g = cm_synth_for(golden_dir, name)
    hp = g["hp"]
    tr = O.OracleCMTrainer(sd, cm_cfg_of(g["cfg"]), g["total_t"], lr=hp["lr"], beta1=hp["beta1"], beta2=hp["beta2"],
                           eps=hp["eps"], weight_decay=hp["weight_decay"], ema_beta=hp["ema_beta"] if hp["ema"] else None,
                           lambda_G=hp["lambda_G"], optim=hp["optim"])
    for it, s in enumerate(g["steps"]):
        loss = tr.optimize_parameters(s["B"], s["mask"], s["noise"], s["timesteps"])
        torch.testing.assert_close(loss, s["loss"], rtol=2e-4, atol=1e-6)
        if "param_checks" in s:
            for k, ref in s["param_checks"].items():
                v = tr.P[k]
                mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
                torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=k)
            if hp["ema"]:
                for k, ref in s["ema_checks"].items():
                    v = tr.ema[k]
                    mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
                    torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=k)


@pytest.mark.parametrize("name", CM_CFGS)
def test_cm_restoration(golden_dir, name):
    """CMGenerator.restoration (multistep consistency sampling, cm_generator.py:504-554) of the unmodified reference with its N(0,1)
    draws recorded (oracle/make_golden_cm.py sampling)"""
    sd, _ = cm_synth_for(golden_dir, name)
    g = load(golden_dir, f"cm_sampling_{name}.pt")
    with torch.no_grad():
        out = O.cm_restoration(sd, g["y_t"], g["mask"], g["sigmas"], g["noises"], cm_cfg_of(g["cfg"]))
    torch.testing.assert_close(out, g["output"], rtol=2e-4, atol=2e-5)
    keep = (g["mask"] == 0).expand_as(out)
    assert torch.equal(out[keep], g["y_t"][keep])          # unmasked pixels are exact copies of the input


# ---- DDPM sampling (restoration): oracle/make_golden_sampling.py fixtures ----------------------------------
@pytest.mark.parametrize("name", ["tiny_eff", "tiny_attn"])
def test_ddpm_restoration(golden_dir, name):
    g = load(golden_dir, f"sampling_{name}.pt")
    ref_sd = {}
    for k in g["keys"]:
        leaf = k.split(".")[-1]
        if O._is_buffer(k):
            ref_sd[k] = g["sched_test"][leaf] if leaf in g["sched_test"] else torch.zeros(g["shapes"][k])
        else:
            ref_sd[k] = torch.empty(g["shapes"][k])
    sd = O.synth_state_dict(ref_sd, seed=0)
    # the short test schedule is the reference's (float64 numpy -> fp32, bit exact)
    mine = O.noise_schedule_buffers("test", g["T"])
    for k, v in g["sched_test"].items():
        assert torch.equal(mine[k], v), k
    with torch.no_grad():
        y, ret = O.ddpm_restoration(sd, g["A"], g["y_t0"], g["B"], g["mask"], g["noises"], cfg_of(g["cfg"]), sample_num=2)
    torch.testing.assert_close(y, g["y_out"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ret, g["ret"], rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        y, ret = O.ddim_restoration(sd, g["A"], g["y_t0"], g["B"], g["mask"], cfg_of(g["cfg"]), sample_num=2, num_steps=4, eta=0.5)
    torch.testing.assert_close(y, g["y_ddim"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ret, g["ret_ddim"], rtol=1e-4, atol=1e-5)


# ---- CUT networks (ResnetGenerator, NLayerDiscriminator): oracle/make_golden_cut.py fixtures ------------------
@pytest.mark.parametrize("name", ["small", "wide"])
def test_cut_networks(golden_dir, name):
    g = load(golden_dir, f"cutnet_{name}.pt")
    c = g["cfg"]
    G, D = g["G"], g["D"]
    P = {k: v.clone().requires_grad_(True) for k, v in O.synth_state_dict({k: torch.empty(G["shapes"][k]) for k in G["keys"]}, 0).items()}
    x = G["x"].clone().requires_grad_(True)
    out = O.resnet_generator(P, x, c["n_blocks"])
    torch.testing.assert_close(out, G["out"], rtol=1e-4, atol=1e-5)
    (out * G["R"]).sum().backward()
    torch.testing.assert_close(x.grad, G["dx"], rtol=1e-3, atol=1e-4)
    for k, ref in G["grad_checks"].items():
        v = P[k].grad
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        torch.testing.assert_close(mine, ref, rtol=1e-3, atol=1e-3 * float(ref[0]) + 1e-5, msg=k)
    with torch.no_grad():
        feats = O.resnet_encoder({k: v.detach() for k, v in P.items()}, G["x"], c["n_blocks"], g["nce_layers"])[1]
    for a, b in zip(feats, G["feats"]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    Pd = {k: v.clone().requires_grad_(True) for k, v in O.synth_state_dict({k: torch.empty(D["shapes"][k]) for k in D["keys"]}, 1).items()}
    xd = G["x"].clone().requires_grad_(True)
    pred = O.nlayer_discriminator(Pd, xd)
    torch.testing.assert_close(pred, D["out"], rtol=1e-4, atol=1e-5)
    (pred * D["R"]).sum().backward()
    torch.testing.assert_close(xd.grad, D["dx"], rtol=1e-3, atol=1e-4)
    for k, ref in D["grad_checks"].items():
        v = Pd[k].grad
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        torch.testing.assert_close(mine, ref, rtol=1e-3, atol=1e-3 * float(ref[0]) + 1e-5, msg=k)


# ---- attention ResNet generators (resnet_attn / mobile_resnet_attn): oracle/make_golden_resattn.py fixtures --------
@pytest.mark.parametrize("name", ["plain", "mobile"])
def test_resnet_attn_generator(golden_dir, name):
    g = load(golden_dir, f"resattn_{name}.pt")
    c = g["cfg"]
    P = {k: v.clone().requires_grad_(True) for k, v in O.synth_state_dict({k: torch.empty(g["shapes"][k]) for k in g["keys"]}, 0).items()}
    x = g["x"].clone().requires_grad_(True)
    out, feats = O.resnet_attn_generator(P, x, c["n_blocks"], g["mobile"], c["nb_mask_attn"], c["nb_mask_input"], g["nce_layers"])
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    assert len(feats) == len(g["feats"]) == 2          # ids beyond the blocks tap nothing
    for a, b in zip(feats, g["feats"]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    (out * g["R"]).sum().backward()
    torch.testing.assert_close(x.grad, g["dx"], rtol=1e-3, atol=1e-4)
    for k, ref in g["grad_checks"].items():
        v = P[k].grad
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        torch.testing.assert_close(mine, ref, rtol=1e-3, atol=1e-3 * float(ref[0]) + 1e-5, msg=k)


# ---- CUT losses and training step: oracle/make_golden_cutstep.py fixtures (unmodified reference modules / CUTModel) --------
class ReplayRandom:
    """replays the recorded python-`random` draws of the reference's image pools, checking the call kinds"""

    def __init__(self, log):
        self.log, self.i = list(log), 0

    def _next(self, kind):
        k, v = self.log[self.i]
        assert k == kind, (self.i, k, kind)
        self.i += 1
        return v

    def uniform(self, a, b):
        return self._next("uniform")

    def randint(self, a, b):
        return self._next("randint")


def _chk(named, refs, rtol, msg="", bias_slack=0.0, flip_slack=0.0):
    """bias_slack: conv biases in front of an InstanceNorm have an analytically zero gradient, so Adam normalises pure rounding
    noise there and moves them by up to lr per step in an arbitrary direction (in the reference too)."""
    for k, ref in refs.items():
        v = named[k].detach()
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        slack = bias_slack * v.numel() ** 0.5 if k.endswith(".bias") else 0.0       # bound on |delta|; its projection: |delta| |pv| ~ slack sqrt(n)
        tol = rtol * float(ref[0]) + rtol * abs(float(ref[1])) + 1e-6
        # flip_slack (= lr * earlier iterations): after the first Adam step two fp32 implementations disagree on the SIGN of the few
        # gradient elements that are rounding noise; each such element ends 2 lr apart.  That leaves the norm alone (it stays tight:
        # a missing or mis-scaled update would show there) but moves the random projection by ~2 lr sqrt(#flipped)
        assert abs(float(mine[0] - ref[0])) <= tol + slack and abs(float(mine[1] - ref[1])) <= tol + (slack + 0.1 * flip_slack) * v.numel() ** 0.5, \
            f"{msg}{k} {mine.tolist()} {ref.tolist()}"


@pytest.mark.parametrize("name", ["a", "b"])
def test_cut_losses(golden_dir, name):
    g = load(golden_dir, f"cutloss_{name}.pt")
    B, P = g["B"], g["P"]
    sdF = O.synth_state_dict({k: torch.empty(g["shapesF"][k]) for k in g["keysF"]}, 3)
    for lname, monce in (("monce", True), ("patchnce", False)):
        r = g[lname]
        Fp = {k: v.clone().requires_grad_(True) for k, v in sdF.items()}
        fk = [f.clone().requires_grad_(True) for f in g["feats_k"]]
        fq = [f.clone().requires_grad_(True) for f in g["feats_q"]]
        k_pool = O.patch_sample_f(Fp, fk, P, g["ids"])
        q_pool = O.patch_sample_f(Fp, fq, P, g["ids"])
        for a, b in zip(k_pool + q_pool, g["k_pool"] + g["q_pool"]):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
        per = [O.patch_nce_loss(q, k, B, 0.07, P, monce) for q, k in zip(q_pool, k_pool)]
        for a, b in zip(per, r["per"]):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
        gq, gk = torch.autograd.grad(per[0].mean(), [q_pool[0], k_pool[0]], retain_graph=True)
        torch.testing.assert_close(gq, r["dq0"], rtol=1e-3, atol=1e-6)
        torch.testing.assert_close(gk, r["dk0"], rtol=1e-3, atol=1e-6)
        total = sum(p.mean() for p in per) / len(per)
        total.backward()
        for a, b in zip([f.grad for f in fq] + [f.grad for f in fk], r["dfeats_q"] + r["dfeats_k"]):
            torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-7)
        _chk({k: v.grad for k, v in Fp.items()}, r["gradF"], 1e-3, lname + " ")
    ls = g["lsgan"]
    pred = ls["pred"].clone().requires_grad_(True)
    l1 = O.lsgan(pred, 1.0)
    torch.testing.assert_close(l1, ls["real"])
    torch.testing.assert_close(torch.autograd.grad(l1, pred)[0], ls["dreal"])
    l0 = O.lsgan(pred, 0.0)
    torch.testing.assert_close(l0, ls["fake"])
    torch.testing.assert_close(torch.autograd.grad(l0, pred)[0], ls["dfake"])


def cut_trainer_for(g):
    c, hp = g["cfg"], g["hp"]
    sdG = O.synth_state_dict({k: torch.empty(g["shapesG"][k]) for k in g["keysG"]}, 0)
    sdD = O.synth_state_dict({k: torch.empty(g["shapesD"][k]) for k in g["keysD"]}, 1)
    sdF = O.synth_state_dict({k: torch.empty(g["shapesF"][k]) for k in g["keysF"]}, 3)
    draws = [d for s in g["steps"] for d in s["pool_draws"]]
    rng = ReplayRandom(draws)
    tr = O.OracleCUTTrainer(sdG, sdF, sdD, c["n_blocks"], [int(i) for i in c["nce_layers"].split(",")], num_patches=c["num_patches"],
                            T=hp["T"], monce=c["nce_loss"] == "monce", lambda_NCE=hp["lambda_NCE"], lambda_GAN=hp["lambda_GAN"],
                            lr_G=hp["lr_G"], lr_D=hp["lr_D"], beta1=hp["beta1"], beta2=hp["beta2"], eps=hp["eps"], pool_size=c["pool"],
                            pool_rng=rng, ema_beta=hp["ema_beta"], gen=cut_gen(c))
    return tr, rng


def cut_gen(c):
    n = c.get("netG", "resnet")
    return "segformer" if "segformer" in n else (n if "resnet_attn" in n else "resnet")


def cut_ntaps(c):
    """number of features get_feats returns: the attention ResNets tap block indices only (resnet_generator.py:504-515)"""
    ids = [int(i) for i in c["nce_layers"].split(",")]
    return len([i for i in ids if 0 <= i < c["n_blocks"]]) if "resnet_attn" in c.get("netG", "") else len(ids)


def cut_ids(step, nlayers, num_patches):
    """the reference draws one randperm per tapped layer on the k pass of each calculate_feats call (cut_networks.py:50-54)"""
    perms = step["perms"]
    assert len(perms) == 2 * nlayers
    ids = [p[: min(num_patches, p.numel())] for p in perms]
    return ids[:nlayers], ids[nlayers:]


@pytest.mark.parametrize("name", ["monce", "patchnce", "config0", "segformer", "mobile_attn"])
def test_cut_steps(golden_dir, name):
    g = load(golden_dir, f"cutstep_{name}.pt")
    c = g["cfg"]
    tr, rng = cut_trainer_for(g)
    nl = cut_ntaps(c)
    for it, s in enumerate(g["steps"]):
        ids_ab, ids_idt = cut_ids(s, nl, c["num_patches"])
        losses = tr.step(s["A"], s["B"], ids_ab, ids_idt, uniforms=s.get("uniforms") or None)
        ref = s["losses"]
        for mine_k, ref_k in (("G_tot", "G_tot"), ("G_GAN", "G_GAN_D_B_basic"), ("G_NCE", "G_NCE"), ("G_NCE_Y", "G_NCE_Y"), ("D_tot", "D_tot")):
            # later iterations inherit the noise-driven Adam steps of the zero-gradient biases (see _chk)
            assert abs(losses[mine_k] - ref[ref_k]) <= 2e-4 * (1 + it) ** 2 * abs(ref[ref_k]) + 1e-5, (it, mine_k, losses[mine_k], ref[ref_k])
        # exact on iteration 0; afterwards the trajectories of two fp32 implementations separate slowly (Adam's sign-like steps on
        # parameters whose gradient is rounding noise: e.g. the key bias of an attention layer, to which softmax is invariant)
        e = float((tr.fake_B - s["fake_B"]).norm() / s["fake_B"].norm())
        assert e < 1e-5 * 30 ** it + 1e-5, (it, e)
        if "G_checks" in s:
            lg, ld = g["hp"]["lr_G"], g["hp"]["lr_D"]
            _chk(tr.G, s["G_checks"], 2e-4, f"G it{it} ", bias_slack=lg * (it + 1), flip_slack=lg * it)
            _chk(tr.Fp, s["F_checks"], 2e-4, f"F it{it} ", flip_slack=lg * it)
            _chk(tr.D, s["D_checks"], 2e-4, f"D it{it} ", bias_slack=ld * (it + 1), flip_slack=ld * it)
            _chk(tr.ema, s["ema_checks"], 2e-4, f"ema it{it} ", bias_slack=lg * (it + 1), flip_slack=lg * it)
    assert rng.i == len(rng.log)


def test_cut_gradient_accumulation(golden_dir):
    """`train_iter_size = 2` on the CUT step (models/base_model.py:1250-1282,1302-1377; the shipped GAN examples train with 8 / 16): four
    calls of the unmodified reference's CUTModel.optimize_parameters() = two optimizer steps of G, F and D (oracle/make_golden_cutaccum.py).
    Pinned per call: the raw losses of both groups, G / F / D unchanged on a non-boundary call and equal to the reference's after a
    boundary, the EMA of G_A updated on EVERY call, the `<name>_avg` values the loss log reports."""
    g = load(golden_dir, "cutstep_accum.pt")
    c, n = g["cfg"], g["iter_size"]
    tr, rng = cut_trainer_for(g)
    nl = cut_ntaps(c)
    names = (("G_tot", "G_tot"), ("G_GAN", "G_GAN_D_B_basic"), ("G_NCE", "G_NCE"), ("G_NCE_Y", "G_NCE_Y"), ("D_tot", "D_tot"))
    lg, ld = g["hp"]["lr_G"], g["hp"]["lr_D"]
    for it, s in enumerate(g["steps"]):
        before = {k: v.clone() for d in (tr.G, tr.Fp, tr.D) for k, v in d.items()}
        ids_ab, ids_idt = cut_ids(s, nl, c["num_patches"])
        losses = tr.iteration(s["A"], s["B"], ids_ab, ids_idt, iter_size=n)
        w = it // n          # optimizer steps taken before this call
        for mine_k, ref_k in names:
            assert abs(losses[mine_k] - s["raw"][ref_k]) <= 2e-4 * (1 + w) ** 2 * abs(s["raw"][ref_k]) + 1e-5, (it, mine_k, losses[mine_k], s["raw"][ref_k])
        if (it + 1) % n:
            assert tr.reported is None and "losses_reported" not in s
            assert all(torch.equal(v, before[k]) for d in (tr.G, tr.Fp, tr.D) for k, v in d.items()), "parameters moved inside a window"
        else:
            for mine_k, ref_k in names:
                ref = s["losses_reported"][ref_k + "_avg"]
                assert abs(tr.reported[mine_k + "_avg"] - ref) <= 2e-4 * (1 + w) ** 2 * abs(ref) + 1e-5, (it, mine_k)
        steps_done = (it + 1) // n
        _chk(tr.G, s["G_checks"], 2e-4, f"G call{it} ", bias_slack=lg * steps_done, flip_slack=lg * max(0, steps_done - 1))
        _chk(tr.Fp, s["F_checks"], 2e-4, f"F call{it} ", flip_slack=lg * max(0, steps_done - 1))
        _chk(tr.D, s["D_checks"], 2e-4, f"D call{it} ", bias_slack=ld * steps_done, flip_slack=ld * max(0, steps_done - 1))
        _chk(tr.ema, s["ema_checks"], 2e-4, f"ema call{it} ", bias_slack=lg * steps_done, flip_slack=lg * max(0, steps_done - 1))
    assert rng.i == len(rng.log)


def test_torch_cpu_instance_norm_channels_last_backward():
    """Documents why oracle._inorm / ref_shim wrap instance_norm: with a channels-last grad_output the stock CPU backward disagrees
    with central finite differences (float64), the wrapped one agrees.  If a future torch fixes the bug both agree -- still green."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 16, 6, 6, dtype=torch.float64, generator=g)
    G = torch.randn(1, 6, 6, 16, dtype=torch.float64, generator=g).permute(0, 3, 1, 2)       # channels-last strides
    orig = getattr(F.instance_norm, "_jg_orig", F.instance_norm)
    xs = x.clone().requires_grad_(True)
    (stock,) = torch.autograd.grad(orig(xs, eps=1e-5), xs, G)
    xo = x.clone().requires_grad_(True)
    (mine,) = torch.autograd.grad(O._inorm(xo), xo, G)
    eps, fd = 1e-6, []
    idx = [(0, 3, 2, 5), (0, 10, 0, 0), (0, 15, 5, 5), (0, 7, 4, 1)]
    for i in idx:
        xp, xm = x.clone(), x.clone()
        xp[i] += eps
        xm[i] -= eps
        fd.append(float(((orig(xp, eps=1e-5) - orig(xm, eps=1e-5)) * G).sum() / (2 * eps)))
    for i, f in zip(idx, fd):
        assert abs(float(mine[i]) - f) < 1e-6 * max(1.0, abs(f)), (i, float(mine[i]), f)
    stock_err = max(abs(float(stock[i]) - f) for i, f in zip(idx, fd))
    print("stock CPU instance_norm backward, channels-last grad: max abs error vs finite differences =", stock_err)


def test_palette_loss_variants(golden_dir):
    """L1 / multiscale_L1 / multiscale_MSE (reference nn.L1Loss and MultiScaleDiffusionLoss called as palette_model.py:597-618 does)"""
    g = load(golden_dir, "palette_loss.pt")
    for key, r in g.items():
        if key[0] == "inputs":
            continue
        S, lossname, use_mask, use_w = key
        inp = O.palette_loss_inputs(S, g[("inputs", S)]["B"])
        assert abs(float(inp["noise_hat"].double().sum()) - g[("inputs", S)]["check"]) < 1e-6
        nh = inp["noise_hat"].clone().requires_grad_(True)
        loss, levels = O.palette_loss_variants(inp["noise"], nh, inp["mask"] if use_mask else None, lossname, inp["w"] if use_w else 1.0)
        torch.testing.assert_close(loss, r["loss"], rtol=1e-5, atol=1e-7)
        assert sorted(levels) == sorted(r["levels"])
        for k, v in r["levels"].items():
            torch.testing.assert_close(levels[k], v, rtol=1e-5, atol=1e-8)
        (gr,) = torch.autograd.grad(loss, nh)
        chk = torch.stack([gr.norm(), (gr * O.projection_vector("palette_loss_grad", gr.shape)).sum()])
        torch.testing.assert_close(chk, r["grad_check"], rtol=1e-4, atol=1e-7)
        if "grad" in r:
            torch.testing.assert_close(gr, r["grad"], rtol=1e-5, atol=1e-9)


# ---- SegFormer attention generator (a19): oracle/make_golden_segformer.py fixtures (unmodified reference, train mode) ----------
@pytest.mark.parametrize("name", ["s64", "s128"])
def test_segformer_generator(golden_dir, name):
    g = load(golden_dir, f"segformer_{name}.pt")
    sd = O.synth_state_dict({k: torch.empty(g["shapes"][k]) for k in g["keys"]}, 4)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = g["x"].clone().requires_grad_(True)
    bn = {}
    out, _ = O.segformer_generator_attn(P, x, rand=iter(g["rands"][: g["n_fwd"]]), bn_state=bn)
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-5)
    for k, v in bn.items():                       # one training forward updated the BatchNorm running statistics
        torch.testing.assert_close(v, g["bn_after"][k], rtol=1e-4, atol=1e-6)
    assert int(g["bn_after"]["final_conv.model.1.num_batches_tracked"]) == 1
    (out * g["R"]).sum().backward()
    torch.testing.assert_close(x.grad, g["dx"], rtol=2e-3, atol=1e-5)
    _chk({k: v.grad for k, v in P.items()}, g["grad_checks"], 2e-3, "segformer ")
    with torch.no_grad():
        feats = O.segformer_backbone(sd, g["x"], rand=iter(g["rands"][g["n_fwd"]:]))
        for a, b in zip(feats, g["feats"]):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
        sd_eval = dict(sd)
        sd_eval.update({k: v for k, v in g["bn_after"].items()})
        out_eval, _ = O.segformer_generator_attn(sd_eval, g["x"], rand=None)
        torch.testing.assert_close(out_eval, g["out_eval"], rtol=1e-4, atol=1e-5)


# ---- projected discriminator (a21 / a24): oracle/make_golden_projd.py fixture ---------------------------------------------------------
def projd_state(g, seed=5):
    return O.synth_state_dict({k: torch.empty(g["shapes"][k]) for k in g["keys"]}, seed=seed)


def projd_run_oracle(P, g):
    """the fixture's sequence: loss_D = (hinge(D(real), real) + hinge(D(fake), fake)) / 2 with its parameter gradients, then the
    generator-side loss -mean(D(fake)) with its gradient to the image; three training forwards = three power iterations"""
    train = [k for k in P if k.startswith("discriminator.") and not (k.endswith("weight_u") or k.endswith("weight_v"))]
    for k in train:
        P[k] = P[k].clone().requires_grad_(True)
    interp = g["cfg"]["interp"]
    pred_real = O.projected_discriminator(P, g["real"], interp)
    pred_fake = O.projected_discriminator(P, g["fake"], interp)
    loss_D = (O.hinge_loss(pred_real, True) + O.hinge_loss(pred_fake, False)) * 0.5
    grads = dict(zip(train, torch.autograd.grad(loss_D, [P[k] for k in train])))
    uv_mid = {k: v.clone() for k, v in P.items() if k.endswith("weight_u") or k.endswith("weight_v")}
    fk = g["fake"].clone().requires_grad_(True)
    loss_G = O.hinge_loss(O.projected_discriminator(P, fk, interp), True, relu=False)
    (dfake,) = torch.autograd.grad(loss_G, fk)
    uv_after = {k: v.clone() for k, v in P.items() if k.endswith("weight_u") or k.endswith("weight_v")}
    return dict(pred_real=pred_real.detach(), loss_D=loss_D.detach(), grads=grads, uv_mid=uv_mid, loss_G=loss_G.detach(), dfake=dfake,
                uv_after=uv_after)


@pytest.mark.parametrize("fixture", ["projd.pt", "projd_lite0.pt"])
def test_projected_discriminator(golden_dir, fixture):
    """the CPU restatement of Proj (CCM / CSM), MultiScaleD / SingleDisc / DownBlock with spectral norm, and the hinge losses against
    the unmodified reference run over the stand-in backbone (projd.pt) and over the tf_efficientnet_lite0 architecture (projd_lite0.pt:
    the reference's own `_make_efficientnet` slicing of oracle/efficientnet_lite0_torch.py, eval-mode BatchNorm, TF SAME padding)"""
    g = load(golden_dir, fixture)
    r = projd_run_oracle(projd_state(g), g)
    torch.testing.assert_close(r["pred_real"], g["pred_real"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(r["loss_D"], g["loss_D"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(r["loss_G"], g["loss_G"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(r["dfake"], g["dfake"], rtol=1e-3, atol=1e-7)
    assert set(r["grads"]) == set(g["grad_checks"])
    for k, ref in g["grad_checks"].items():
        v = r["grads"][k]
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        torch.testing.assert_close(mine, ref, rtol=2e-4, atol=2e-4 * float(ref[0]) + 1e-7, msg=k)
    for name in ("uv_mid", "uv_after"):
        for k, ref in g[name].items():
            v = r[name][k]
            mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
            torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-5, msg=(name, k))


@pytest.mark.parametrize("fixture", ["projd_vit.pt", "projd_vit256.pt"])
def test_projected_discriminator_vit(golden_dir, fixture):
    """`D_proj_network_type = "vitsmall"` (examples/example_gan_mario2sonic.json, BASELINE configs[2]): the CPU restatement of the ViT token
    path (`configure_get_feats_vit_timm`), the Conv1d CCM, the FeatureFusionBlockVector CSM and the MLP heads of MultiScaleD(conv=False)
    against the unmodified reference run over oracle/vit_small_torch.py (37 tokens at interp 96, 257 tokens at interp 256)"""
    g = load(golden_dir, fixture)
    P = projd_state(g)
    with torch.no_grad():
        feats = O.projd_features(P, torch.nn.functional.interpolate(g["real"], g["cfg"]["interp"], mode="bilinear", align_corners=False))
    for i, f in enumerate(feats):
        v = f.transpose(1, 2)
        assert tuple(v.shape) == g["feat_shapes"][str(i)]
        mine = torch.stack([v.norm(), (v * O.projection_vector(str(i), v.shape)).sum()])
        torch.testing.assert_close(mine, g["feat_checks"][str(i)], rtol=1e-4, atol=1e-4 * float(g["feat_checks"][str(i)][0]) + 1e-6, msg=str(i))
    r = projd_run_oracle(P, g)
    torch.testing.assert_close(r["pred_real"], g["pred_real"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(r["loss_D"], g["loss_D"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(r["loss_G"], g["loss_G"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(r["dfake"], g["dfake"], rtol=1e-3, atol=1e-6 * float(g["dfake"].abs().max()))
    assert set(r["grads"]) == set(g["grad_checks"]) and len(r["grads"]) == 24
    for k, ref in g["grad_checks"].items():
        v = r["grads"][k]
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        torch.testing.assert_close(mine, ref, rtol=2e-4, atol=2e-4 * float(ref[0]) + 1e-7, msg=k)


# ---- rounding yardstick: fp32 reference arithmetic with 16-bit storage between layers ----------------------------------------
YARD_C2 = dict(ngf=64, mults=[1, 2, 4, 8], res_blocks=[2, 2, 2, 2], attn_res=[16], efficient=True, S=256, B=1)
YARD_MED = dict(ngf=32, mults=[1, 2, 4], res_blocks=[1, 1, 1], attn_res=[16], efficient=True, S=64, B=2)
YARD_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_rounding_yardstick.json")


def rounding_yardstick(c, seed=5):
    """first-step loss / noise_hat / per-parameter gradient distance between the plain fp32 oracle and the same oracle with every
    inter-layer activation, every back-propagated activation gradient and the weights rounded to fp16 / bf16 (oracle/jg_oracle.py
    `activation_rounding`): what a perfect 16-bit execution of this network is allowed to differ by.  Same inputs, same error
    measure (incl. the analytic-zero bias floor) as tests/test_gpu_1_model.py::_first_step_vs_oracle."""
    from joligen_amd.models.palette_model import define_G
    from joligen_amd.options import opt_from_json

    ov = dict(G_ngf=c["ngf"], G_unet_mha_channel_mults=c["mults"], G_unet_mha_res_blocks=c["res_blocks"], G_unet_mha_attn_res=c["attn_res"],
              G_unet_mha_vit_efficient=c["efficient"], data_crop_size=c["S"], train_batch_size=c["B"])
    net = define_G(**vars(opt_from_json({}, ov)))
    sd = O.synth_state_dict(net.state_dict(), seed=0)
    cfg = O.UNetCfg(in_channel=6, inner_channel=c["ngf"], out_channel=3, res_blocks=c["res_blocks"], attn_res=c["attn_res"],
                    channel_mults=c["mults"], efficient=c["efficient"])
    B, S = c["B"], c["S"]
    g = torch.Generator().manual_seed(seed)
    Bimg = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    mask = torch.zeros(B, 1, S, S, dtype=torch.int64)
    mask[:, :, S // 6:(5 * S) // 8, S // 3:(7 * S) // 9] = 1
    A = Bimg * (1 - mask) + torch.randn(B, 3, S, S, generator=g) * mask
    t, u, noise = O.draw_step_randomness(torch.Generator().manual_seed(9), Bimg, 2000)
    tr = O.OraclePaletteTrainer(sd, cfg, ema_beta=None)
    loss_ref, grads_ref, nh_ref = tr.loss_and_grads(Bimg, A, mask, noise, t, u)
    ref_norms = {k: float(v.norm()) for k, v in grads_ref.items()}
    out = {}
    for name, dt, eps16 in (("fp16", torch.float16, 2.0 ** -10), ("bf16", torch.bfloat16, 2.0 ** -7)):
        sd16 = {k: (v.to(dt).float() if (torch.is_floating_point(v) and v.dim() >= 3) else v) for k, v in sd.items()}   # conv weights: 16-bit copies
        tr16 = O.OraclePaletteTrainer(sd16, cfg, ema_beta=None)
        tr16.grad_scale = 65536.0 if dt == torch.float16 else 1.0       # the static fp16 loss scale of the HIP path (base_model.py)
        with O.activation_rounding(dt):
            loss, grads, nh = tr16.loss_and_grads(Bimg, A, mask, noise, t, u)
        errs = []
        for k, gr in grads_ref.items():
            floor = 0.0
            if k.endswith(".bias") and (k[:-4] + "weight") in ref_norms:
                floor = 16.0 * eps16 * ref_norms[k[:-4] + "weight"]
            errs.append(float((grads[k] - gr).norm() / (gr.norm() + floor + 1e-12)))
        errs.sort()
        out[name] = dict(loss_rel=abs(float(loss) - float(loss_ref)) / float(loss_ref),
                         noise_hat_rel=float((nh - nh_ref).norm() / nh_ref.norm()),
                         grad_median=errs[len(errs) // 2], grad_p90=errs[int(len(errs) * 0.9)], grad_worst=errs[-1])
    return out


def test_rounding_yardstick():
    """Measured rounding floor of the BASELINE shape (C2, batch 1) and of the 64x64 test shape; the committed file is what the GPU
    tolerances are compared with (tests/test_gpu_1_model.py::test_tolerances_vs_rounding_yardstick).  Regenerate with
    JG_WRITE_YARDSTICK=1."""
    import json

    res = {"c2_256": rounding_yardstick(YARD_C2), "medium_64": rounding_yardstick(YARD_MED)}
    for shape in res.values():
        assert shape["fp16"]["grad_median"] < shape["bf16"]["grad_median"]        # 3 more mantissa bits
        assert 0 < shape["fp16"]["noise_hat_rel"] < 1e-2 and 0 < shape["bf16"]["noise_hat_rel"] < 1e-1
    if os.environ.get("JG_WRITE_YARDSTICK"):
        res["_meta"] = "oracle/jg_oracle.py activation_rounding(dtype): fp32 arithmetic, 16-bit storage of activations, activation gradients and conv weights"
        with open(YARD_FILE, "w") as f:
            json.dump(res, f, indent=1)
    committed = json.load(open(YARD_FILE))
    for shape, r in res.items():
        if shape.startswith("_"):
            continue
        for dt in ("fp16", "bf16"):
            for k, v in r[dt].items():
                assert abs(v - committed[shape][dt][k]) <= 0.25 * committed[shape][dt][k] + 1e-6, (shape, dt, k, v, committed[shape][dt][k])


# ---- rounding yardstick of the CUT step: what a perfect 16-bit execution of G / F / D may differ by -----------
YARD_CUT_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_rounding_yardstick_cut.json")
YARD_CUT_CASES = ["monce", "segformer", "mobile_attn"]


def cut_zero_grad_bias(gen):
    """parameters of the CUT generators whose gradient is analytically zero (a conv bias in front of an InstanceNorm, the key bias
    of an attention layer): compared with an absolute floor taken from the same layer's weight gradient"""
    def skip(k):
        if gen == "segformer":
            return k.endswith("in_proj_bias")
        if "resnet_attn" in gen:
            return k.endswith(".bias") and not k.startswith("deconv3_")
        return k.endswith(".bias")
    return skip


def cut_first_step_oracle(g, dtype16, rounded):
    """first G-group (and D-group) backward of the oracle CUT trainer on the step-0 inputs of a cutstep fixture, weights and inputs
    made `dtype16`-representable exactly as tests/test_gpu_5_cutloss.py::test_cut_model_first_step_gradients_vs_oracle does;
    `rounded`: with every inter-layer activation (and its gradient) of G and D stored in dtype16 (oracle `activation_rounding`) and the
    HIP path's static fp16 loss scale of the CUT model (1024, models/cut_model.py)."""
    import random

    c = g["cfg"]
    s = g["steps"][0]
    tr, _ = cut_trainer_for(g)
    isbuf = lambda k: "running_" in k or "num_batches_tracked" in k
    r16 = lambda v: v.to(dtype16).float() if torch.is_floating_point(v) else v
    sdG = {k: r16(v) for k, v in O.synth_state_dict({k: torch.empty(g["shapesG"][k]) for k in g["keysG"]}, 0).items()}
    tr.G = {k: v.clone() for k, v in sdG.items() if not isbuf(k)}
    tr.Gbuf = {k: v.clone() for k, v in sdG.items() if isbuf(k)}
    tr.D = {k: r16(v) for k, v in tr.D.items()}
    tr.pool.rng = tr.real_pools[0].rng = tr.real_pools[1].rng = random.Random(0)
    nl = cut_ntaps(c)
    ids_ab, ids_idt = cut_ids(s, nl, c["num_patches"])
    A, Bi = r16(s["A"]), r16(s["B"])
    if rounded:
        tr.grad_scale = 1024.0 if dtype16 == torch.float16 else 1.0
        with O.activation_rounding(dtype16):
            losses = tr.step(A, Bi, ids_ab, ids_idt, uniforms=s.get("uniforms") or None)
    else:
        losses = tr.step(A, Bi, ids_ab, ids_idt, uniforms=s.get("uniforms") or None)
    return tr, losses


def cut_rounding_yardstick(golden_dir, name):
    g = load(golden_dir, f"cutstep_{name}.pt")
    gen = cut_gen(g["cfg"])
    skip = cut_zero_grad_bias(gen)
    out = {}
    for tag, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        ref, lref = cut_first_step_oracle(g, dt, rounded=False)
        rnd, lrnd = cut_first_step_oracle(g, dt, rounded=True)
        res = {"loss_rel": {k: abs(lrnd[k] - lref[k]) / (abs(lref[k]) + 1e-12) for k in lref},
               "fake_B_rel": float((rnd.fake_B - ref.fake_B).norm() / ref.fake_B.norm())}
        # PatchGAN: the biases of the convolutions that feed an InstanceNorm (all but the first and the last one) have a zero gradient
        dconv = sorted(int(k.split(".")[1]) for k in ref.last_grads["D"] if k.endswith(".bias"))
        dzero = {f"model.{i}.bias" for i in dconv[1:-1]}
        for key in ("G", "F", "D"):
            errs, coss = [], []
            for k, gr in ref.last_grads[key].items():
                mine = rnd.last_grads[key][k]
                floor = 0.0
                if (key == "G" and skip(k)) or (key == "D" and k in dzero):
                    wk = k.replace("in_proj_bias", "in_proj_weight") if k.endswith("in_proj_bias") else k[:-4] + "weight"
                    floor = 2e-3 * float(ref.last_grads[key][wk].norm())
                errs.append(float((mine.double() - gr.double()).norm()) / (float(gr.norm()) + floor + 1e-30))
                if float(gr.norm()) > 10 * floor and float(gr.norm()) > 1e-6:
                    coss.append(float((mine.double().flatten() @ gr.double().flatten()) / (float(mine.double().norm()) * float(gr.double().norm()) + 1e-300)))
            errs.sort()
            res[key] = dict(grad_median=errs[len(errs) // 2], grad_p90=errs[int(len(errs) * 0.9)], grad_worst=errs[-1], cos_min=min(coss))
        out[tag] = res
    return out


@pytest.mark.parametrize("name", YARD_CUT_CASES)
def test_cut_rounding_yardstick(golden_dir, name):
    """Measured rounding floor of the first CUT step (G / F / D gradients, losses, fake_B) for the fixtures the GPU gradient tests use;
    the committed numbers bound tests/test_gpu_5_cutloss.py::test_cut_model_first_step_gradients_vs_oracle.  JG_WRITE_YARDSTICK=1 rewrites
    the file (one entry per case)."""
    import json

    res = cut_rounding_yardstick(golden_dir, name)
    for key in ("G", "F", "D"):
        assert res["fp16"][key]["grad_median"] <= res["bf16"][key]["grad_median"] * 1.05 + 1e-9
        assert res["fp16"][key]["grad_worst"] > 0
    committed = json.load(open(YARD_CUT_FILE)) if os.path.exists(YARD_CUT_FILE) else {}
    if os.environ.get("JG_WRITE_YARDSTICK"):
        committed[name] = res
        committed["_meta"] = ("oracle/jg_oracle.py OracleCUTTrainer.step on cutstep_<case>.pt step 0 under activation_rounding(dtype): fp32 arithmetic, "
                              "16-bit storage of every inter-layer activation and activation gradient of G and D; 16-bit-representable weights and inputs on both sides")
        with open(YARD_CUT_FILE, "w") as f:
            json.dump(committed, f, indent=1)
    for dt in ("fp16", "bf16"):
        for key in ("G", "F", "D"):
            for k, v in res[dt][key].items():
                c = committed[name][dt][key][k]
                if k == "cos_min":
                    assert v >= c - 0.05, (name, dt, key, k, v, c)
                else:      # the comparison is chaotic (ReLU masks flip under rounding): the statistic itself moves with the torch build
                    assert 0.5 * c - 1e-6 <= v <= 2.0 * c + 1e-6, (name, dt, key, k, v, c)


# ---- rounding yardstick of the projected discriminator (both backbones) ---------------------------------------------------------------
YARD_PROJD_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_rounding_yardstick_projd.json")


def projd_rounding_yardstick(golden_dir, fixture):
    """the fixture's D step + G-side image gradient in the fp32 oracle against the same oracle with 16-bit storage of every inter-layer
    activation / activation gradient (`activation_rounding`), weights and inputs 16-bit-representable on both sides: ReLU6 / LeakyReLU
    masks that flip under rounding make the image gradient through the 16 frozen MBConv blocks (and the small second mini-discriminator's
    gradients) far noisier than the forward -- this measures by how much"""
    g = load(golden_dir, fixture)
    out = {}
    for tag, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        r16 = lambda v: v.to(dt).float()
        gg = dict(g, real=r16(g["real"]), fake=r16(g["fake"]))
        P0 = {k: (r16(v) if (torch.is_floating_point(v) and not k.endswith(("weight_u", "weight_v"))) else v) for k, v in projd_state(g).items()}
        ref = projd_run_oracle({k: v.clone() for k, v in P0.items()}, gg)
        with O.activation_rounding(dt):
            rnd = projd_run_oracle({k: v.clone() for k, v in P0.items()}, gg)
        errs = sorted(float((rnd["grads"][k] - ref["grads"][k]).norm() / (ref["grads"][k].norm() + 1e-30)) for k in ref["grads"])
        out[tag] = dict(pred_real_rel=float((rnd["pred_real"] - ref["pred_real"]).norm() / ref["pred_real"].norm()),
                        dfake_rel=float((rnd["dfake"] - ref["dfake"]).norm() / ref["dfake"].norm()),
                        grad_median=errs[len(errs) // 2], grad_worst=errs[-1])
        if "vit" in fixture:
            # round 5: distance of the 16-bit run to the FIXTURE itself (fp32 weights and inputs): rounding the weights on the way in moves the
            # MLP heads' pre-activations, and ONE ReLU of the 2 x 400 that flips changes the image gradient of that sample by several per cent
            out[tag]["dfake_rel_fixture"] = float((rnd["dfake"] - g["dfake"]).norm() / g["dfake"].norm())
    return out


@pytest.mark.parametrize("fixture", ["projd.pt", "projd_lite0.pt", "projd_vit.pt", "projd_vit256.pt"])
def test_projd_rounding_yardstick(golden_dir, fixture):
    """committed floor of tests/test_gpu_6_projd.py's gradient tolerances (JG_WRITE_YARDSTICK=1 rewrites the entry)"""
    import json

    res = projd_rounding_yardstick(golden_dir, fixture)
    committed = json.load(open(YARD_PROJD_FILE)) if os.path.exists(YARD_PROJD_FILE) else {}
    if os.environ.get("JG_WRITE_YARDSTICK"):
        committed[fixture] = res
        committed["_meta"] = "oracle projd_run_oracle under activation_rounding(dtype) vs plain fp32, 16-bit-representable weights / inputs on both sides"
        with open(YARD_PROJD_FILE, "w") as f:
            json.dump(committed, f, indent=1)
    for dt in ("fp16", "bf16"):
        for k, v in res[dt].items():
            c = committed[fixture][dt][k]
            assert 0.5 * c - 1e-6 <= v <= 2.0 * c + 1e-6, (fixture, dt, k, v, c)


# ---- class-conditioned palette_model (alg_diffusion_cond_embed = "class"): oracle/make_golden_cond.py fixture -------------------
def cls_state(g, T_test=None):
    sched = {}
    ref_sd = {}
    for k in g["keys"]:
        leaf = k.split(".")[-1]
        if O._is_buffer(k):
            phase = "train" if leaf.endswith("_train") else "test"
            n = g["shapes"][k][0]
            if (phase, n) not in sched:
                sched[(phase, n)] = O.noise_schedule_buffers(phase, n)
            ref_sd[k] = sched[(phase, n)][leaf]
        else:
            ref_sd[k] = torch.empty(g["shapes"][k])
    sd = O.synth_state_dict(ref_sd, seed=0)
    row, scale = g["table_row_scale"]
    sd[g["table_key"]][row] *= scale      # a row longer than max_norm
    return sd


def palette_conditioning_dropout(drop_u, p, num_classes, cls, mask):
    """compute_palette_loss (palette_model.py:565-584): for the dropped samples BOTH conditionings are replaced by the highest class --
    the class label, and every pixel of the mask (which, clamped to [0, 1] downstream, makes the whole image of that sample "masked")."""
    drop = drop_u < p
    if cls is not None:
        cls = torch.where(drop, torch.full_like(cls, num_classes - 1), cls)
    mask = torch.where(drop.reshape(-1, 1, 1, 1).expand(mask.shape), torch.full_like(mask, num_classes - 1), mask)
    return cls, mask


@pytest.mark.parametrize("tag", ["cls", "mask"])
def test_palette_conditioning(golden_dir, tag):
    g = load(golden_dir, f"palette_{tag}_tiny.pt")
    c = g["cfg"]
    cfg = cfg_of(c)
    assert g["num_classes"] == c["nclasses"] + 1            # the unconditioned class of the conditioning dropout
    # generator forward with the conditioning; the looked-up row longer than max_norm is renormalised in place
    sd = cls_state(g)
    f = g["fwd"]
    k = g["table_key"]
    assert float(sd[k][2].norm()) > 1.0
    with torch.no_grad():
        _, noise_hat, w, _ = O.diffusion_generator_forward(sd, f["B"], f["A"], f["mask"], f["noise"], f["t"], f["u"], cfg, cls=f["cls"])
    torch.testing.assert_close(noise_hat, f["noise_hat"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(sd[k][2].norm(), f["table_row2_norm_after"], rtol=1e-5, atol=1e-6)
    assert float(sd[k][2].norm()) <= 1.0 + 1e-5
    # three optimizer steps with the conditioning dropout
    sd = cls_state(g)
    hp = g["hp"]
    tr = O.OraclePaletteTrainer(sd, cfg, lr=hp["lr"], beta1=hp["beta1"], beta2=hp["beta2"], eps=hp["eps"], weight_decay=hp["weight_decay"],
                                ema_beta=hp["ema_beta"], lambda_G=hp["lambda_G"], optim=hp["optim"])
    for it, s in enumerate(g["steps"]):
        cls, mask = palette_conditioning_dropout(s["drop_u"], c["dropout_prob"], g["num_classes"], s["cls"], s["mask"])
        loss = tr.optimize_parameters(s["B"], s["A"], mask, s["noise"], s["t"], s["u"], cls=cls)
        torch.testing.assert_close(loss, s["loss"], rtol=2e-4, atol=1e-6)
        if "param_checks" in s:
            for kk, ref in s["param_checks"].items():
                v = tr.P[kk]
                mine = torch.stack([v.norm(), (v * O.projection_vector(kk, v.shape)).sum()])
                torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=kk)
            for kk, ref in s["ema_checks"].items():
                v = tr.ema[kk]
                mine = torch.stack([v.norm(), (v * O.projection_vector(kk, v.shape)).sum()])
                torch.testing.assert_close(mine, ref, rtol=1e-4, atol=1e-4 * float(ref[0]) + 1e-6, msg=kk)
    # sampling with the conditioning
    sd = cls_state(g)
    sm = g["sampling"]
    with torch.no_grad():
        y, ret = O.ddpm_restoration(sd, sm["A"], sm["y_t0"], sm["B"], sm["mask"], sm["noises"], cfg, sample_num=2, cls=sm["cls"])
    torch.testing.assert_close(y, sm["y_out"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ret, sm["ret"], rtol=1e-4, atol=1e-5)


# ---- input transforms (f3): the resize restatement against Pillow's own outputs -------------------------------------------------------
def test_pil_resize_restatement(golden_dir):
    """oracle/pil_resize.py (Pillow's fixed-point BICUBIC resampler and its incrementally-accumulated NEAREST, restated) against
    `Image.resize` outputs committed by oracle/make_golden_resize.py: bit-exact, down- / up-sampling, non-square"""
    import pil_resize as R

    g = load(golden_dir, "resize_pil.pt")
    assert len(g["cases"]) >= 5
    for c in g["cases"]:
        oh, ow = c["out_hw"]
        assert torch.equal(torch.from_numpy(R.resize_bicubic_u8(c["img"].numpy(), oh, ow)), c["img_resized"]), (tuple(c["img"].shape), oh, ow)
        assert torch.equal(torch.from_numpy(R.resize_nearest_u8(c["mask"].numpy(), oh, ow)), c["mask_resized"]), (tuple(c["mask"].shape), oh, ow)
    # the product's host-side tables are the same numbers
    from joligen_amd.data_device import pil_bicubic_tables, pil_nearest_table

    for a, b in ((56, 32), (29, 48), (64, 24), (96, 71)):
        bo, kk = R.precompute_coeffs(a, b)
        pb, pk = pil_bicubic_tables(a, b)
        assert torch.equal(pb, torch.from_numpy(bo)) and torch.equal(pk, torch.from_numpy(kk))
        assert torch.equal(pil_nearest_table(a, b).long(), torch.from_numpy(R.nearest_index_table(a, b)))


# ---- the committed fixtures regenerate from the committed recipes ------------------------------------------------
def test_every_recipe_is_registered():
    import regen_check as RC

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    on_disk = sorted(f for f in os.listdir(os.path.join(root, "oracle")) if f.startswith("make_golden") and f.endswith(".py"))
    assert sorted(RC.RECIPES) == on_disk


def test_fixture_hashes_match_the_regeneration_record(golden_dir):
    """tests/golden/REGENERATED.json (written by `python oracle/regen_check.py --write` after a full bit-for-bit regeneration) lists
    exactly the committed fixtures with their hashes -- runs everywhere, also where the reference tree is absent."""
    import json

    import regen_check as RC

    rec = json.load(open(RC.MANIFEST))
    assert rec["fixtures"] == RC.fixture_digests(golden_dir)


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="the reference tree is only present in the build container")
def test_fixtures_regenerate(golden_dir, tmp_path):
    """Every file under tests/golden/ is an output of the UNMODIFIED reference: run every oracle/make_golden*.py (they import
    /root/reference through oracle/ref_shim.py) into a scratch directory and require each regenerated fixture to equal the committed
    one bit for bit -- a recipe that drifts away from the fixture it once wrote (as make_golden_cutstep.py did in round 2) fails here.

    The full run is ~7 minutes of CPU and its outcome is a function of its inputs (oracle/*.py, the reference's *.py, torch / numpy
    versions): it runs when the digest of those inputs differs from the one recorded by the last full run (oracle/regen_check.py), or
    with JG_FULL_REGEN=1; with unchanged inputs the recorded fixture hashes are compared instead."""
    import json

    import regen_check as RC

    rec = json.load(open(RC.MANIFEST)) if os.path.exists(RC.MANIFEST) else {}
    if os.environ.get("JG_FULL_REGEN", "0") == "0" and rec.get("inputs_sha256") == RC.inputs_digest():
        assert rec["fixtures"] == RC.fixture_digests(golden_dir)
        return
    RC.full_check(str(tmp_path))

"""CPU: the oracle restatement (oracle/jg_oracle.py) against fixtures produced by the
unmodified reference (oracle/make_golden.py).  This is what pins the oracle."""
import os

import pytest
import torch

import jg_oracle as O

CFGS = ["tiny_eff", "tiny_noeff", "tiny_attn"]


def cfg_of(c):
    return O.UNetCfg(in_channel=6, inner_channel=c["ngf"], out_channel=3, res_blocks=c["res_blocks"],
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

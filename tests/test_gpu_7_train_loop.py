"""GPU: the reference's training loop (train.py:195-301, 351-357) replayed against joligen_amd with the SHIPPED example configurations
(tests/examples/*.json are verbatim copies of /root/reference/examples/example_ddpm_noglasses2glasses.json,
example_gan_horse2zebra.json and example_gan_mario2sonic.json) plus the measurement overrides of SURVEY.md Appendix C (the examples' own train_iter_size 16 / 8 is kept:
two accumulation windows each):

    opt = parse(example JSON + overrides) -> create_model -> data_dependent_initialize -> setup -> single_gpu
    2 x iter_size x (set_input, optimize_parameters) + get_current_losses -> save_networks("latest") -> export_networks("latest")
    -> update_learning_rate -> a second process-like model continues from the checkpoint (train_continue)

What it pins: the drop-in boundary b1 -- the example JSONs load unchanged, the model API is the one
train.py drives, the checkpoints carry reference-layout keys, and `export_networks` no longer ends the loop for `cut`."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
EX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "examples")


def _loop(opt, data, n_windows):
    """train.py:194-357 without the dataloader / visualizer / metrics: n_windows optimizer steps = n_windows * train_iter_size calls of
    optimize_parameters() (models/base_model.py:1250-1282: gradients accumulate over a window, the optimizers step at its end).  Checked
    on the way: the parameters of every network stay put inside a window and move at its boundary."""
    from joligen_amd.models import create_model

    model = create_model(opt, 0)
    if hasattr(model, "data_dependent_initialize"):
        model.data_dependent_initialize(data)
    model.setup(opt)
    model.single_gpu()
    n = opt.train_iter_size
    losses = []

    def fingerprint():
        return {name: float(model._net(name).arena.p.double().sum()) for name in model.model_names}

    for w in range(n_windows):
        start = fingerprint()
        for j in range(n):
            model.set_input(data)
            model.optimize_parameters()
            now = fingerprint()
            if j < n - 1:
                assert now == start, (w, j, "parameters moved inside an accumulation window")
            else:
                assert all(now[k] != start[k] for k in now), (w, "a network did not step at the window boundary", now, start)
        losses.append({k: float(v) for k, v in model.get_current_losses().items()})       # output_print_freq path (:288-303)
    return model, losses


def _appendix_c(tmp_path, **kw):
    """the measurement overrides of SURVEY.md Appendix C -- WITHOUT train_iter_size: the examples keep their own (16 / 8)"""
    ov = dict(output_display_type=["none"], output_print_freq=10 ** 9, checkpoints_dir=str(tmp_path), gpu_ids="0",
              train_metrics_list=[], jg_act_dtype="bf16")
    ov.update(kw)
    return ov


def test_example_ddpm_json_through_the_train_loop(tmp_path):
    from joligen_amd.options import opt_from_json
    from bench import synth_batch

    ov = _appendix_c(tmp_path, name="ddpm_e2e", data_crop_size=128, data_load_size=128, train_batch_size=4)
    opt = opt_from_json(os.path.join(EX, "example_ddpm_noglasses2glasses.json"), ov)
    assert opt.model_type == "palette" and opt.G_netG == "unet_mha" and opt.train_G_ema and opt.train_optim == "adamw"
    data = synth_batch(4, 128, 3, torch.device("cuda:0"))
    torch.manual_seed(0)
    assert opt.train_iter_size == 16                      # the JSON's own accumulation window
    model, losses = _loop(opt, data, 2)
    assert all(math.isfinite(v) for l in losses for v in l.values()) and "G_tot_avg" in losses[0]
    assert losses[-1]["G_tot_avg"] < losses[0]["G_tot_avg"] * 1.5
    model.save_networks("latest")
    assert model.export_networks("latest") == []            # the reference skips palette / cm as well (base_model.py:885-891)
    lr0 = model.optimizers[0].param_groups[0]["lr"]
    model.update_learning_rate()
    assert model.optimizers[0].param_groups[0]["lr"] <= lr0
    sd = torch.load(os.path.join(str(tmp_path), "ddpm_e2e", "latest_net_G_A.pth"), map_location="cpu")
    assert "denoise_fn.model.input_blocks.0.0.weight" in sd and sd["denoise_fn.model.input_blocks.0.0.weight"].shape == (64, 6, 3, 3)
    assert os.path.exists(os.path.join(str(tmp_path), "ddpm_e2e", "latest_net_G_A_ema.pth"))
    # continue from the checkpoint: the reloaded parameters are the saved ones
    opt2 = opt_from_json(os.path.join(EX, "example_ddpm_noglasses2glasses.json"), dict(ov, train_continue=True))
    from joligen_amd.models import create_model

    m2 = create_model(opt2, 0)
    m2.setup(opt2)
    sd2 = m2._net("G_A").state_dict()
    for k, v in sd.items():
        assert torch.equal(sd2[k].cpu(), v), k


def test_example_gan_horse2zebra_json_through_the_train_loop(tmp_path):
    """mobile_resnet_attn generator + [projected_d, basic] discriminators, MoNCE, 256x256, batch 4 -- the JSON as shipped"""
    from joligen_amd.options import opt_from_json

    ov = _appendix_c(tmp_path, name="h2z_e2e", train_export_jit=True)
    opt = opt_from_json(os.path.join(EX, "example_gan_horse2zebra.json"), ov)
    assert opt.model_type == "cut" and opt.G_netG == "mobile_resnet_attn" and opt.D_netDs == ["projected_d", "basic"] and opt.train_batch_size == 4
    g = torch.Generator().manual_seed(5)
    data = {"A": torch.rand(4, 3, 256, 256, generator=g) * 2 - 1, "B": torch.rand(4, 3, 256, 256, generator=g) * 2 - 1,
            "A_img_paths": ["synthetic"] * 4, "B_img_paths": ["synthetic"] * 4}
    torch.manual_seed(0)
    assert opt.train_iter_size == 8                       # the JSON's own accumulation window
    model, losses = _loop(opt, data, 2)
    assert set(losses[0]) >= {k + "_avg" for k in ("G_tot", "G_NCE", "G_NCE_Y", "G_GAN_D_B_projected_d", "G_GAN_D_B_basic", "D_tot")}
    assert all(math.isfinite(v) for l in losses for v in l.values()), losses
    model.save_networks("latest")
    written = model.export_networks("latest")          # train.py:352,357: called after EVERY save
    d = os.path.join(str(tmp_path), "h2z_e2e")
    assert os.path.join(d, "latest_net_G_A.pt") in written          # TorchScript (train_export_jit); ONNX needs the `onnx` package
    model.update_learning_rate()
    # the exported graph reproduces the HIP generator on the same input (fp32 CPU trace of the same weights vs bf16 kernels)
    jit = torch.jit.load(os.path.join(d, "latest_net_G_A.pt"))
    x = data["A"][:1]
    with torch.no_grad():
        ref = jit(x)
    model.netG_A.eval()
    from joligen_amd import ops

    with torch.no_grad():
        y = model.netG_A(ops.to_nhwc(x.to("cuda:0"), torch.bfloat16, 8))
    y = y.permute(0, 3, 1, 2)[:, :3].float().cpu()
    assert float((y - ref).norm() / ref.norm()) < 5e-2
    for name in ("G_A", "F", "D_B_projected_d", "D_B_basic"):
        assert os.path.exists(os.path.join(d, f"latest_net_{name}.pth")), name
    sd = torch.load(os.path.join(d, "latest_net_G_A.pth"), map_location="cpu")
    assert "resnet_blocks.0.conv1.conv.0.weight" in sd and "deconv3_attention.weight" in sd


def test_example_gan_mario2sonic_json_through_the_train_loop(tmp_path):
    """BASELINE configs[2]'s own JSON (examples/example_gan_mario2sonic.json, verbatim): segformer_attn_conv generator, projected discriminator
    with the ViT projector (`proj_network_type: "vitsmall"`, `proj_interp: 256` -> 257 tokens) + basic, MoNCE, lsgan / hinge, crop 128, batch 2.
    Two overrides for branches that are out of scope: `vision_aided` dropped from D_netDs (CLIP / DINO / Swin backbones) and
    the semantic-mask branch (f_s network + online mask dataset) off."""
    from joligen_amd.options import opt_from_json

    ov = _appendix_c(tmp_path, name="m2s_e2e", D_netDs=["projected_d", "basic"], train_semantic_mask=False, train_mask_out_mask=False)
    opt = opt_from_json(os.path.join(EX, "example_gan_mario2sonic.json"), ov)
    assert opt.model_type == "cut" and opt.G_netG == "segformer_attn_conv" and opt.D_proj_network_type == "vitsmall" and opt.D_proj_interp == 256
    assert opt.train_batch_size == 2 and opt.data_crop_size == 128 and opt.train_iter_size == 1
    g = torch.Generator().manual_seed(6)
    data = {"A": torch.rand(2, 3, 128, 128, generator=g) * 2 - 1, "B": torch.rand(2, 3, 128, 128, generator=g) * 2 - 1,
            "A_img_paths": ["synthetic"] * 2, "B_img_paths": ["synthetic"] * 2}
    torch.manual_seed(0)
    model, losses = _loop(opt, data, 4)
    from joligen_amd.modules.projected_d_vit import MultiScaleDVit, ProjVit

    netD = model.netD_B_projected_d
    assert isinstance(netD.freeze_feature_network, ProjVit) and isinstance(netD.discriminator, MultiScaleDVit)
    assert netD.freeze_feature_network.RESOLUTIONS == [257] * 4
    assert set(losses[0]) >= {"G_tot", "G_NCE", "G_NCE_Y", "G_GAN_D_B_projected_d", "G_GAN_D_B_basic", "D_tot", "D_GAN_D_B_projected_d"}
    assert all(math.isfinite(v) for l in losses for v in l.values()), losses
    model.save_networks("latest")
    sd = torch.load(os.path.join(str(tmp_path), "m2s_e2e", "latest_net_D_B_projected_d.pth"), map_location="cpu")
    assert sd["freeze_feature_network.pretrained.pos_embed"].shape == (1, 257, 384)
    assert sd["freeze_feature_network.scratch.layer3_ccm.weight"].shape == (512, 384, 1)
    assert sd["discriminator.mini_discs.3.1.weight"].shape == (100, 256 * 257)


def test_exchange_path_on_one_gpu_costs_under_a_millisecond():
    """multi-GPU readiness without a multi-GPU box: `bench.py --gpus 1 --force-exchange` runs the data-parallel step --
    FlatDataParallel, EarlyExchange chunks launched from inside the backward, RCCL all-reduce on a 1-rank group, chunk-pipelined fused
    AdamW -- at BASELINE configs[1]'s shape; against the plain single-GPU step of the same process layout it may cost < 1 ms per step
    (the exchange of a 1-rank group moves no data: what is measured is the launch structure the 8-GPU run will use)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "4", "--no-cpu-baseline", "--no-cut-leg", "--no-kernel-timing"]

    def run(extra):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    plain, exch = run([]), run(["--force-exchange"])
    assert exch["config"]["n_ranks_seen"] == 1 and "exchange" in exch
    assert exch["ms_per_step_median"] - plain["ms_per_step_median"] < 1.0, (plain["ms_per_step_median"], exch["ms_per_step_median"])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/force_exchange_vs_dp1.json", "w") as f:
        json.dump({"dp1_ms_median": plain["ms_per_step_median"], "force_exchange_ms_median": exch["ms_per_step_median"], "exchange": exch["exchange"]}, f)


def test_cut_exchange_path_on_one_gpu_keeps_the_fast_driver():
    """VERDICT r4 missing #3: the data-parallel CUT step (FlatDataParallel, chunked RCCL all-reduce of the four gradient arenas issued by their
    optimizer steps, on a 1-rank group) runs the SAME step driver as the single-GPU step -- discriminator half on the side stream, replayed from
    its hipGraph -- at the BASELINE configs[2] shape, and may cost < 1.5 ms per step against the plain step of the same process layout."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--model", "cut", "--netG", "segformer_attn_conv", "--netDs", "projected_d,basic", "--proj", "vitsmall",
            "--batch", "16", "--steps", "12", "--warmup", "5", "--no-cpu-baseline", "--no-kernel-timing"]

    def run(extra):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    plain, exch = run([]), run(["--force-exchange"])
    import joligen_amd

    want = "graph" if joligen_amd.HIP_GRAPHS_SAFE else "early"
    assert plain["config"]["step_driver"].startswith(want), plain["config"]
    assert exch["config"]["step_driver"].startswith(want) and "exchange" in exch, exch["config"]
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/cut_force_exchange_vs_dp1.json", "w") as f:
        json.dump({"dp1": {k: plain[k] for k in ("value", "ms_per_step", "ms_per_step_median")}, "dp1_driver": plain["config"]["step_driver"],
                   "force_exchange": {k: exch[k] for k in ("value", "ms_per_step", "ms_per_step_median")}, "force_exchange_driver": exch["config"]["step_driver"],
                   "exchange": exch["exchange"]}, f)
    assert exch["ms_per_step_median"] - plain["ms_per_step_median"] < 1.5, (plain["ms_per_step_median"], exch["ms_per_step_median"])

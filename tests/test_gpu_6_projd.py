"""GPU parity tests of the projected discriminator: spectral-norm convolution, bilinear (align_corners) resize,
hinge objective, and the whole ProjectedDiscriminator (stand-in backbone -> CCM -> CSM -> four mini-discriminators) against the fixture
recorded from the unmodified reference (oracle/make_golden_projd.py) and against the CPU oracle."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

import jg_oracle as O
from test_oracle_golden import projd_run_oracle, projd_state

pytestmark = pytest.mark.gpu
D0 = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("hw,out", [((7, 9), (14, 18)), ((64, 64), (96, 96)), ((10, 13), (25, 17)), ((1, 5), (4, 5))])
def test_bilinear_align_corners(dtype, align, hw, out):
    """round 5: non-integer ratios as well (the `F.interpolate(x, D_proj_interp)` in front of the projected discriminator with e.g. 64 -> 96:
    the backward's gather window was only right for integer ratios)"""
    from joligen_amd.modules.projected_d import bilinear

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 24, *hw, generator=g).to(dtype)
    gy = torch.randn(2, 24, *out, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    yr = F.interpolate(xr, size=out, mode="bilinear", align_corners=align)
    yr.backward(gy.float())
    xd = x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True)
    y = bilinear(xd, out[0], out[1], align)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(D0))
    assert relerr(y.permute(0, 3, 1, 2), yr.detach()) < TOL[dtype]
    assert relerr(xd.grad.permute(0, 3, 1, 2), xr.grad) < 2 * TOL[dtype], relerr(xd.grad.permute(0, 3, 1, 2), xr.grad)


@pytest.mark.parametrize("dtype", DTYPES)
def test_hinge_loss(dtype):
    from joligen_amd.modules.projected_d import hinge_loss

    g = torch.Generator().manual_seed(2)
    p = (torch.randn(3, 100, generator=g) * 1.5).to(dtype)
    for real, relu in ((True, True), (False, True), (True, False)):
        pr = p.float().requires_grad_(True)
        lo = O.hinge_loss(pr, real, relu)
        (lo * 3.0).backward()
        pd = p.to(D0).requires_grad_(True)
        l = hinge_loss(pd, real, relu)
        (l * 3.0).backward()
        assert abs(float(l) - float(lo)) < 1e-5 * abs(float(lo)) + 1e-6
        assert relerr(pd.grad, pr.grad) < 2e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,stride,pad,bias", [(64, 128, 4, 2, 1, True), (256, 1, 4, 1, 0, False)])
def test_spectral_conv_vs_torch(cin, cout, k, stride, pad, bias, dtype):
    """spectral_norm(nn.Conv2d) of blocks.py:11-13: two training forwards (two power iterations, each with its own sigma) followed by
    one backward through both -- output, input gradient, gradient w.r.t. weight_orig (through 1 / sigma) and bias, and the updated
    u / v, against torch.nn.utils.spectral_norm on the rounded inputs."""
    import torch.nn as nn

    from joligen_amd.arena import ParamArena
    from joligen_amd.modules.projected_d import SpectralConv2d

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.c = SpectralConv2d(cin, cout, k, stride, pad, bias=bias)

    torch.manual_seed(3)
    m = M()
    ref = torch.nn.utils.spectral_norm(nn.Conv2d(cin, cout, k, stride, pad, bias=bias))
    with torch.no_grad():
        ref.weight_orig.copy_(m.c.weight_orig.to(dtype).float())
        m.c.weight_orig.copy_(ref.weight_orig)
        ref.weight_u.copy_(m.c.weight_u)
        ref.weight_v.copy_(m.c.weight_v)
        if bias:
            ref.bias.copy_(m.c.bias)
    ParamArena(m, D0, dtype, priority=()).refresh()
    m.train()
    ref.train()
    g = torch.Generator().manual_seed(4)
    S = 16
    xs = [torch.randn(2, cin, S, S, generator=g).to(dtype) for _ in range(2)]
    outs, outs_r, xds, xrs = [], [], [], []
    for x in xs:
        xd = x.permute(0, 2, 3, 1).contiguous().to(D0).requires_grad_(True)
        xr = x.float().requires_grad_(True)
        outs.append(m.c(xd))
        outs_r.append(ref(xr))
        xds.append(xd)
        xrs.append(xr)
    gys = [torch.randn(outs_r[0].shape, generator=g).to(dtype) for _ in range(2)]
    sum((o * gy.float()).sum() for o, gy in zip(outs_r, gys)).backward()
    cp = outs[0].shape[-1]
    for o, gy in zip(outs, gys):
        gyp = torch.zeros(gy.shape[0], gy.shape[2], gy.shape[3], cp, dtype=dtype)
        gyp[..., :cout] = gy.permute(0, 2, 3, 1)
        o.backward(gyp.to(D0))
    torch.cuda.synchronize()
    for o, orf in zip(outs, outs_r):
        assert relerr(o.permute(0, 3, 1, 2)[:, :cout], orf.detach()) < TOL[dtype]
    for xd, xr in zip(xds, xrs):
        assert relerr(xd.grad.permute(0, 3, 1, 2), xr.grad) < 2 * TOL[dtype], relerr(xd.grad.permute(0, 3, 1, 2), xr.grad)
    assert relerr(m.c.weight_orig.grad, ref.weight_orig.grad) < 2 * TOL[dtype], relerr(m.c.weight_orig.grad, ref.weight_orig.grad)
    if bias:
        assert relerr(m.c.bias.grad, ref.bias.grad) < 2 * TOL[dtype]
    assert relerr(m.c.weight_u, ref.weight_u) < 1e-5 and relerr(m.c.weight_v, ref.weight_v) < 1e-5


FIXTURES = ["projd.pt", "projd_lite0.pt"]     # stand-in backbone (round 2) | tf_efficientnet_lite0 architecture (round 3)


def _yard(fixture, dtype):
    """measured rounding floor of this fixture's gradients (tests/test_oracle_golden.py::test_projd_rounding_yardstick)"""
    import json

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_rounding_yardstick_projd.json")
    return json.load(open(path))[fixture]["fp16" if dtype == torch.float16 else "bf16"]


def _backbone_of(g):
    """which backbone drove the reference when the fixture was written: timm's key names (conv_dw / conv_pwl) = tf_efficientnet_lite0"""
    return "lite0" if any("conv_dw" in k for k in g["keys"]) else "standin"


def build_projd(g, dtype, seed=5):
    import warnings

    from joligen_amd.modules.projected_d import ProjectedDiscriminator

    c = g["cfg"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")       # "random frozen weights" / "stand-in": the fixture's synthetic weights are loaded right below
        net = ProjectedDiscriminator("efficientnet", interp=c["interp"], img_size=c["S"], backbone=_backbone_of(g))
    assert list(net.state_dict().keys()) == g["keys"]
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(g["shapes"][k]), k
    net.load_state_dict(projd_state(g, seed))
    net.jg_finalize(torch.device(D0), dtype)
    net.train()
    return net


@pytest.mark.parametrize("fixture", FIXTURES)
@pytest.mark.parametrize("dtype", DTYPES)
def test_projected_discriminator_vs_reference_golden(golden_dir, dtype, fixture):
    """The fixture's sequence on HIP: D(real), D(fake) (training forwards: spectral-norm power iterations), the hinge discriminator
    loss and the gradient of all 44 trainable tensors, then the generator-side loss and its gradient w.r.t. the fake image through the
    frozen feature network (backbone, CCM, CSM).  state_dict keys / shapes are the reference's.  projd_lite0.pt: the reference ran over
    the tf_efficientnet_lite0 ARCHITECTURE (16 MBConv blocks, TF SAME padding, eval-mode BatchNorm, ReLU6; synthetic weights -- parity of
    the backbone against timm itself is unpinned, timm is absent), the HIP side runs csrc/effnet.hip + the MFMA 1x1 convolutions."""
    from joligen_amd import ops
    from joligen_amd.modules.projected_d import hinge_loss

    g = load(golden_dir, fixture)
    net = build_projd(g, dtype)
    for n, p in net.named_parameters():
        p.requires_grad_(not n.startswith("freeze"))
    real = ops.to_nhwc(g["real"].to(D0), dtype, 8)
    fake = ops.to_nhwc(g["fake"].to(D0), dtype, 8)
    pred_real = net(real)
    pred_fake = net(fake)
    assert tuple(pred_real.shape) == tuple(g["pred_real"].shape)
    tol = 6e-3 if dtype == torch.float16 else 4e-2
    assert relerr(pred_real, g["pred_real"]) < tol, relerr(pred_real, g["pred_real"])
    loss_D = (hinge_loss(pred_real, True) + hinge_loss(pred_fake, False)) * 0.5
    assert abs(float(loss_D) - float(g["loss_D"])) < tol * abs(float(g["loss_D"]))
    net.arena.g.zero_()
    loss_D.backward()
    torch.cuda.synchronize()
    bad = []
    P = dict(net.named_parameters())
    for k, ref in g["grad_checks"].items():
        v = P[k].grad.detach().float().cpu()
        mine = torch.stack([v.norm(), (v * O.projection_vector(k, v.shape)).sum()])
        # GroupNorm over groups of TWO channels + LeakyReLU + a hinge (sign-like) loss: the first blocks' gradients carry 4 - 6 % of
        # 16-bit rounding noise against the fp32 reference (per-parameter table of the oracle test below), the last ones 0.1 %
        t = max(4 * tol, 0.08, 2.0 * _yard(fixture, dtype)["grad_worst"]) * float(ref[0]) + 1e-7      # >= twice the measured rounding floor
        if abs(float(mine[0] - ref[0])) > t or abs(float(mine[1] - ref[1])) > 2 * t * max(1.0, v.numel() ** 0.5 / 4):
            bad.append((k, mine.tolist(), ref.tolist()))
    assert not bad, bad[:6]
    sd = net.state_dict()
    for k, ref in g["uv_mid"].items():
        v = sd[k].float().cpu()
        assert abs(float(v.norm()) - float(ref[0])) < 1e-4 and abs(float((v * O.projection_vector(k, v.shape)).sum()) - float(ref[1])) < 5e-3, k
    # generator side: gradient to the image through the frozen feature network (no parameter gradient is produced there)
    fk = ops.to_nhwc(g["fake"].to(D0), dtype, 8).requires_grad_(True)
    loss_G = hinge_loss(net(fk), True, relu=False)
    assert abs(float(loss_G) - float(g["loss_G"])) < tol * abs(float(g["loss_G"])) + 2e-3
    # fp16: the image gradient of this loss is ~1e-6 per pixel -- below fp16's normal range after 16 frozen blocks; the model applies its
    # static loss scale to exactly this backward (models/cut_model.py: 1024 for fp16)
    ls = 1024.0 if dtype == torch.float16 else 1.0
    (loss_G * ls).backward()
    dfk = fk.grad.permute(0, 3, 1, 2)[:, :3].float() / ls
    # ReLU6 / LeakyReLU masks flip under 16-bit rounding: the image gradient through the frozen network is bounded by twice its
    # measured rounding floor (0.03 / 0.10 stand-in, 0.076 / 0.26 tf_efficientnet_lite0 for fp16 / bf16)
    assert relerr(dfk, g["dfake"]) < 2.0 * _yard(fixture, dtype)["dfake_rel"], (relerr(dfk, g["dfake"]), _yard(fixture, dtype)["dfake_rel"])
    for n, p in net.named_parameters():
        if n.startswith("freeze"):
            assert not p.requires_grad


@pytest.mark.parametrize("fixture", FIXTURES)
def test_projected_discriminator_first_step_vs_oracle(golden_dir, fixture):
    """per-parameter gradients (not just checksums) of the discriminator loss against the CPU oracle on fp16-representable weights"""
    from joligen_amd import ops
    from joligen_amd.modules.projected_d import hinge_loss

    dtype = torch.float16
    g = load(golden_dir, fixture)
    g = dict(g, real=g["real"].half().float(), fake=g["fake"].half().float())
    P0 = {k: (v.half().float() if (torch.is_floating_point(v) and not k.endswith(("weight_u", "weight_v"))) else v) for k, v in projd_state(g).items()}
    import warnings

    from joligen_amd.modules.projected_d import ProjectedDiscriminator

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = ProjectedDiscriminator("efficientnet", interp=g["cfg"]["interp"], img_size=g["cfg"]["S"],
                                     backbone=_backbone_of(g))
    net.load_state_dict(P0)
    net.jg_finalize(torch.device(D0), dtype)
    net.train()
    for n, p in net.named_parameters():
        p.requires_grad_(not n.startswith("freeze"))
    r = projd_run_oracle({k: v.clone() for k, v in P0.items()}, g)
    real, fake = ops.to_nhwc(g["real"].to(D0), dtype, 8), ops.to_nhwc(g["fake"].to(D0), dtype, 8)
    loss_D = (hinge_loss(net(real), True) + hinge_loss(net(fake), False)) * 0.5
    net.arena.g.zero_()
    loss_D.backward()
    torch.cuda.synchronize()
    assert abs(float(loss_D) - float(r["loss_D"])) < 6e-3 * abs(float(r["loss_D"]))
    errs = []
    for k, p in net.named_parameters():
        if k in r["grads"]:
            errs.append((relerr(p.grad, r["grads"][k]), k))
    errs.sort(reverse=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/grad_table_{fixture[:-3]}.txt", "w") as f:
        f.write("\n".join(f"{e:10.3e} {k}" for e, k in errs))
    y = _yard(fixture, dtype)
    assert errs[0][0] < max(0.1, 2.0 * y["grad_worst"]), (errs[:6], y)
    assert errs[len(errs) // 2][0] < max(0.03, 1.5 * y["grad_median"]), (errs[len(errs) // 2], y)


def test_cut_model_with_projected_and_basic_discriminators():
    """BASELINE configs[2]'s discriminator set through the model API (D_netDs = [projected_d, basic], the example_gan_*.json choice):
    the step runs, every loss the reference logs exists and is finite, the projected discriminator trains with the hinge objective,
    its frozen feature network does not move, both discriminators and the generator do."""
    from joligen_amd.models import create_model
    from joligen_amd.options import opt_from_json

    cfg = {"model_type": "cut", "G": {"netG": "resnet", "ngf": 32, "nblocks": 2}, "D": {"netDs": ["projected_d", "basic"], "ndf": 32, "proj_interp": 128},
           "alg": {"cut": {"nce_layers": "0,4,8"}},
           "data": {"crop_size": 64, "load_size": 64}, "train": {"batch_size": 2, "G_ema": True}}
    model = create_model(opt_from_json(cfg, overrides={"jg_act_dtype": "bf16", "gpu_ids": "0"}), 0)
    g = torch.Generator().manual_seed(2)
    data = {"A": torch.rand(2, 3, 64, 64, generator=g) * 2 - 1, "B": torch.rand(2, 3, 64, 64, generator=g) * 2 - 1}
    torch.manual_seed(0)
    model.data_dependent_initialize(data)
    assert model.discriminators_names == ["D_B_projected_d", "D_B_basic"]
    assert model.D_B_projected_d_loss_calculator.gan_mode == "projected" and model.D_B_basic_loss_calculator.gan_mode == "lsgan"
    before = {n: {k: p.detach().clone() for k, p in getattr(model, "net" + n).named_parameters()} for n in ("G_A", "D_B_projected_d", "D_B_basic")}
    for _ in range(2):
        model.set_input(data)
        model.optimize_parameters()
    torch.cuda.synchronize()
    losses = {k: float(v) for k, v in model.get_current_losses().items()}
    assert set(losses) == {"G_tot", "G_NCE", "G_NCE_Y", "G_GAN_D_B_projected_d", "G_GAN_D_B_basic", "D_tot", "D_GAN_D_B_projected_d", "D_GAN_D_B_basic"}
    assert all(math.isfinite(v) for v in losses.values()), losses
    assert abs(losses["D_tot"] - losses["D_GAN_D_B_projected_d"] - losses["D_GAN_D_B_basic"]) < 1e-3 * abs(losses["D_tot"]) + 1e-4
    for n, params in before.items():
        cur = dict(getattr(model, "net" + n).named_parameters())
        moved = [k for k, p0 in params.items() if not torch.equal(cur[k].detach(), p0)]
        frozen = [k for k in params if k.startswith("freeze")]
        assert all(k not in moved for k in frozen), [k for k in frozen if k in moved][:3]
        assert len(moved) >= 0.8 * (len(params) - len(frozen)), (n, len(moved), len(params))


def _timm_style_lite0_state_dict(seed=11):
    """a state_dict with timm's `tf_efficientnet_lite0` key names and shapes (what `timm.create_model(...).state_dict()` holds: stem,
    bn1, blocks.<stage>.<block>.*, and the classification tail the projector never uses), filled with seeded non-trivial values
    (BatchNorm running statistics included).  timm itself is absent offline: the names come from its published definition restated in
    oracle/efficientnet_lite0_torch.py."""
    from efficientnet_lite0_torch import TfEfficientNetLite0

    net = TfEfficientNetLite0().eval()
    g = torch.Generator().manual_seed(seed)
    sd = net.state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(1234)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) * 0.8 + 0.4
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.2
        elif v.dim() == 1:       # BatchNorm weight / bias
            sd[k] = torch.randn(v.shape, generator=g) * 0.2 + (1.0 if k.endswith("weight") else 0.0)
        else:
            fan = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / fan) ** 0.5
    net.load_state_dict(sd)
    full = dict(sd)
    full.update({"conv_head.weight": torch.randn(1280, 320, 1, 1, generator=g), "bn2.weight": torch.ones(1280), "bn2.bias": torch.zeros(1280),
                 "bn2.running_mean": torch.zeros(1280), "bn2.running_var": torch.ones(1280), "bn2.num_batches_tracked": torch.tensor(0),
                 "classifier.weight": torch.randn(1000, 1280, generator=g) * 0.01, "classifier.bias": torch.zeros(1000)})
    return net, sd, full


@pytest.mark.parametrize("dtype", DTYPES)
def test_pretrained_backbone_loads_and_matches_the_torch_mirror(tmp_path, dtype):
    """f2: `jg_projd_pretrained` -> ProjectedDiscriminator.load_pretrained_backbone with a timm-keyed
    `tf_efficientnet_lite0` state_dict (/root/reference/models/modules/projected_d/projector.py:51-59,251-255 loads timm's weights and
    re-homes the modules into layer0..3): every backbone tensor is consumed, the classification tail is ignored, the frozen BatchNorm
    statistics are folded as the kernels expect, and the four stage features of the HIP feature network equal the plain-torch mirror's
    on the same 16-bit-representable input."""
    import warnings

    from joligen_amd import ops
    from joligen_amd.modules.projected_d import ProjectedDiscriminator

    mirror, sd_backbone, sd_full = _timm_style_lite0_state_dict()
    path = os.path.join(str(tmp_path), "tf_efficientnet_lite0.pth")
    torch.save(sd_full, path)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        net = ProjectedDiscriminator("efficientnet", interp=-1, img_size=64, pretrained_path=path)
    assert net.backbone_pretrained and not [w for w in wlist if "RANDOM frozen weights" in str(w.message)]
    pre = net.freeze_feature_network.pretrained.state_dict()
    assert len(pre) == len(sd_backbone)                      # every timm backbone entry has a home, nothing else lives there
    # re-homing as in _make_efficientnet: layer0 = conv_stem + bn1 + act + blocks[0:2], layer1 = blocks[2], layer2 = blocks[3:5], layer3 = blocks[5:]
    assert torch.equal(pre["layer0.0.weight"], sd_backbone["conv_stem.weight"]) and torch.equal(pre["layer0.1.running_var"], sd_backbone["bn1.running_var"])
    assert torch.equal(pre["layer1.0.0.conv_dw.weight"], sd_backbone["blocks.2.0.conv_dw.weight"])
    assert torch.equal(pre["layer3.1.0.bn3.running_mean"], sd_backbone["blocks.6.0.bn3.running_mean"])
    net.jg_finalize(torch.device(D0), dtype)
    net.eval()
    g = torch.Generator().manual_seed(3)
    img = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).to(dtype).float()
    x = ops.to_nhwc(img.to(D0), dtype, 8)
    p = net.freeze_feature_network.pretrained
    def mirror_stages(rounded):
        """the four stage outputs of the torch mirror; rounded: every convolution / BatchNorm(+ReLU6) / block output is stored in `dtype`
        (where the HIP network keeps 16-bit tensors) -- the rounding yardstick of THIS random network, measured on the spot"""
        hooks = []
        if rounded:
            rnd = lambda m, i, o: o.to(dtype).float()
            for m in mirror.modules():
                if isinstance(m, (torch.nn.Conv2d, torch.nn.BatchNorm2d)) or type(m).__name__ in ("DepthwiseSeparableConv", "InvertedResidual"):
                    hooks.append(m.register_forward_hook(rnd))
        try:
            h, outs = mirror.bn1(mirror.conv_stem(img)), []
            for i, stage in enumerate(mirror.blocks):
                h = stage(h)
                if i in (1, 2, 4, 6):
                    outs.append(h)
            return outs
        finally:
            for hk in hooks:
                hk.remove()

    with torch.no_grad():
        net.arena.ensure_fresh()
        o0 = p.layer0(x); o1 = p.layer1(o0); o2 = p.layer2(o1); o3 = p.layer3(o2)
        refs, floor = mirror_stages(False), mirror_stages(True)
    for i, (mine, ref, fl) in enumerate(zip((o0, o1, o2, o3), refs, floor)):
        assert mine.shape[-1] == ref.shape[1] and mine.shape[1] == ref.shape[2], (i, mine.shape, ref.shape)
        e, ef = relerr(mine.permute(0, 3, 1, 2).float(), ref), relerr(fl, ref)
        # a wrong BatchNorm fold / tap alignment / stage cut is an O(1) error; 16-bit storage through 16 random ReLU6 layers costs `ef`
        assert e < 2.0 * ef + 2e-3, (i, e, ef)
    # a checkpoint WITHOUT one backbone tensor must not load
    bad = {k: v for k, v in sd_full.items() if k != "blocks.3.1.conv_pw.weight"}
    torch.save(bad, path)
    with pytest.raises(RuntimeError):
        ProjectedDiscriminator("efficientnet", interp=-1, img_size=64, pretrained_path=path)


def test_reference_layout_discriminator_checkpoint_restores_the_backbone(tmp_path):
    """a `<epoch>_net_D_B_projected_d.pth` in the reference's layout (376 keys incl. `freeze_feature_network.pretrained.*`, what the
    reference writes with timm's weights inside) through BaseModel.load_networks: every backbone tensor arrives, the model knows its
    feature network is no longer random; the same file without the backbone is refused."""
    import warnings

    from joligen_amd.models import create_model
    from joligen_amd.options import opt_from_json

    cfg = {"model_type": "cut", "G": {"netG": "resnet", "ngf": 16, "nblocks": 2}, "D": {"netDs": ["projected_d"], "proj_interp": 64},
           "alg": {"cut": {"nce_layers": "0,4,8"}}, "data": {"crop_size": 64, "load_size": 64}, "train": {"batch_size": 1}}
    ov = {"jg_act_dtype": "bf16", "gpu_ids": "0", "checkpoints_dir": str(tmp_path), "name": "pd"}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = create_model(opt_from_json(cfg, overrides=ov), 0)
    g = torch.Generator().manual_seed(4)
    data = {"A": torch.rand(1, 3, 64, 64, generator=g) * 2 - 1, "B": torch.rand(1, 3, 64, 64, generator=g) * 2 - 1}
    model.data_dependent_initialize(data)
    model.single_gpu()
    netD = model.netD_B_projected_d
    assert not netD.backbone_pretrained
    _, sd_backbone, _ = _timm_style_lite0_state_dict(seed=21)
    ck = {k: v.detach().cpu().clone() for k, v in netD.state_dict().items()}
    pre = "freeze_feature_network.pretrained."
    n_pre = len([k for k in ck if k.startswith(pre)])
    assert n_pre == len(sd_backbone)
    probe = {}
    for k in list(ck):
        if k.startswith(pre) and torch.is_floating_point(ck[k]) and ck[k].dim() > 0:
            ck[k] = torch.randn(ck[k].shape, generator=g) * 0.1 + (1.0 if k.endswith("running_var") else 0.0)
            probe[k] = ck[k]
    model.save_networks("latest")
    torch.save(ck, os.path.join(str(tmp_path), "pd", "latest_net_D_B_projected_d.pth"))
    model.load_networks("latest")
    now = netD.state_dict()
    for k, v in probe.items():
        assert torch.equal(now[k].cpu(), v), k
    assert netD.backbone_pretrained
    torch.save({k: v for k, v in ck.items() if not k.startswith(pre)}, os.path.join(str(tmp_path), "pd", "latest_net_D_B_projected_d.pth"))
    with pytest.raises(RuntimeError):
        model.load_networks("latest")


# ---- tf_efficientnet_lite0 kernels in isolation: depth-wise k x k + frozen-BatchNorm affine + ReLU6, TF "SAME" padding -------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,stride,H,W", [(3, 1, 12, 12), (3, 2, 13, 11), (5, 1, 9, 14), (5, 2, 14, 15), (5, 2, 16, 16), (3, 2, 7, 7)])
def test_dwconv_affine_relu6_vs_torch(k, stride, H, W, dtype):
    """jg_dwconv_affine_act_fwd / _bwd against F.conv2d(groups = C) on TF-SAME padded input, per-channel affine, ReLU6 (forward and input
    gradient) on 16-bit-representable inputs: a wrong tap alignment under the asymmetric stride-2 padding, or a wrong pass-through set of
    the ReLU6 backward, is an O(1) error here -- the end-to-end fixture bounds it only through 16 layers of rounding."""
    from joligen_amd import _lib, ops
    from joligen_amd.modules.projected_d import tf_same_pad

    L = _lib.lib()
    B, C = 2, 24
    g = torch.Generator().manual_seed(k * 100 + stride * 10 + H)
    x = (torch.randn(B, C, H, W, generator=g) * 2).to(dtype).float()
    w = torch.randn(C, 1, k, k, generator=g) * 0.4
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    (pt, pb), (pl, pr) = tf_same_pad(H, k, stride), tf_same_pad(W, k, stride)
    xr = x.clone().requires_grad_(True)
    pre = F.conv2d(F.pad(xr, (pl, pr, pt, pb)), w, stride=stride, groups=C) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    yr = pre.clamp(0.0, 6.0)
    Ho, Wo = yr.shape[2], yr.shape[3]
    gy = torch.randn(yr.shape, generator=g).to(dtype).float()
    # keep the comparison away from the two kinks: pre-activations within rounding distance of 0 or 6 may pass the gradient on one side only
    safe = ((pre.detach().abs() > 0.05) & ((pre.detach() - 6.0).abs() > 0.05)).float()
    (yr * gy * safe).sum().backward()
    dt = _lib.JG_F16 if dtype == torch.float16 else _lib.JG_BF16
    xd = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(D0)
    wd = w.view(C, k * k).t().contiguous().to(D0)              # [k*k][C]
    sc, sh = scale.to(D0), shift.to(D0)
    y = torch.empty(B, Ho, Wo, C, device=D0, dtype=dtype)
    _lib.check(L.jg_dwconv_affine_act_fwd(dt, xd.data_ptr(), wd.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), B, H, W, C, k, stride, pt, pl,
                                          Ho, Wo, 1, ops._st()))
    tol = 2e-3 if dtype == torch.float16 else 1e-2
    assert relerr(y.permute(0, 3, 1, 2).float(), yr.detach()) < tol
    dyd = (gy * safe).permute(0, 2, 3, 1).contiguous().to(dtype).to(D0)
    dx = torch.empty_like(xd)
    _lib.check(L.jg_dwconv_affine_act_bwd(dt, dyd.data_ptr(), y.data_ptr(), wd.data_ptr(), sc.data_ptr(), dx.data_ptr(), B, H, W, C, k, stride, pt, pl,
                                          Ho, Wo, 1, ops._st()))
    torch.cuda.synchronize()
    assert relerr(dx.permute(0, 3, 1, 2).float(), xr.grad) < 2 * tol, relerr(dx.permute(0, 3, 1, 2).float(), xr.grad)


@pytest.mark.parametrize("dtype", DTYPES)
def test_chan_affine_relu6_vs_torch(dtype):
    from joligen_amd import _lib, ops

    L = _lib.lib()
    P, C = 300, 40
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(P, C, generator=g) * 3).to(dtype).float()
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xr = x.clone().requires_grad_(True)
    pre = xr * scale + shift
    yr = pre.clamp(0.0, 6.0)
    gy = torch.randn(P, C, generator=g).to(dtype).float()
    safe = ((pre.detach().abs() > 0.05) & ((pre.detach() - 6.0).abs() > 0.05)).float()
    (yr * gy * safe).sum().backward()
    dt = _lib.JG_F16 if dtype == torch.float16 else _lib.JG_BF16
    xd, sc, sh = x.to(dtype).to(D0), scale.to(D0), shift.to(D0)
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    _lib.check(L.jg_chan_affine_act_fwd(dt, xd.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), P, C, 1, ops._st()))
    dyd = (gy * safe).to(dtype).to(D0)
    _lib.check(L.jg_chan_affine_act_bwd(dt, dyd.data_ptr(), y.data_ptr(), sc.data_ptr(), dx.data_ptr(), P, C, 1, ops._st()))
    torch.cuda.synchronize()
    tol = 2e-3 if dtype == torch.float16 else 1e-2
    assert relerr(y.float(), yr.detach()) < tol and relerr(dx.float(), xr.grad) < 2 * tol

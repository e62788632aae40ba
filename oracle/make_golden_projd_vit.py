"""Generate tests/golden/projd_vit.pt and projd_vit256.pt by running the UNMODIFIED reference ProjectedDiscriminator("vitsmall") (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container:   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_projd_vit.py

`D_proj_network_type = "vitsmall"` is what examples/example_gan_mario2sonic.json (BASELINE configs[2]) selects.  timm and its pretrained
`vit_small_patch16_224` weights are not available offline: `timm.create_model` is stubbed to return oracle/vit_small_torch.py::VitSmallPatch16
(timm's architecture restated with timm's attribute names; weights synthetic, "backbone parity unpinned").  Everything else is the reference's
own code: `configure_get_feats_vit_timm` (projector.py:138-153: tokens behind blocks 2 / 5 / 8 / 11, transposed to [B, C, T]), `calc_channels`,
the Conv1d CCM and the FeatureFusionBlockVector CSM (projector.py:455-487, blocks.py:290-320), `MultiScaleD(conv=False)` -- four
Flatten / Linear / ReLU MLPs on [B, C * T] (discriminator.py:208-230) --, `ProjectedDiscriminator.forward` with its bilinear resize to
`interp`, the hinge objective of GANLoss("projected") and DiscriminatorGANLoss.compute_loss_D / compute_loss_G.

Two fixtures: projd_vit.pt (64 x 64 images resized to 96: 37 tokens, ragged against every tile size) and projd_vit256.pt (resized to 256:
257 tokens, the `proj_interp` of the shipped example).  Pinned: logits of D(real), the discriminator loss and the gradient of every trainable
parameter (the MLP heads; the projector is frozen), the generator-side loss and its gradient with respect to the fake image.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
from make_golden import checks  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate


def main(interp=96, out_name="projd_vit.pt", S=64, B=2):
    import timm

    from vit_small_torch import VitSmallPatch16

    calls = []

    def create_model(name, img_size=224, pretrained=False, **kw):
        calls.append((name, img_size))
        return VitSmallPatch16(img_size)

    timm.create_model = create_model
    from models.modules.loss import DiscriminatorGANLoss
    from models.modules.projected_d.discriminator import ProjectedDiscriminator

    torch.manual_seed(0)
    netD = ProjectedDiscriminator("vitsmall", interp=interp, img_size=S)
    assert calls == [("vit_small_patch16_224", interp)], calls
    ref_sd = netD.state_dict()
    netD.load_state_dict(O.synth_state_dict(ref_sd, seed=5))
    netD.train()
    g = torch.Generator().manual_seed(78)
    real = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    fake = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    lossf = DiscriminatorGANLoss(netD=netD, device=torch.device("cpu"), dataaug_APA_p=0, dataaug_APA_target=0.6, train_batch_size=B,
                                 dataaug_APA_nimg=50, dataaug_APA_every=4, dataaug_D_label_smooth=False, train_gan_mode="projected",
                                 dataaug_APA=False, dataaug_D_diffusion=False, dataaug_D_diffusion_every=4)
    for p in netD.discriminator.parameters():
        p.requires_grad_(True)
    loss_D = lossf.compute_loss_D(netD, real, fake, None)
    pred_real = lossf.pred_real.detach().clone()
    loss_D.backward()
    grads = {k: p.grad.detach().clone() for k, p in netD.named_parameters() if p.grad is not None}
    fk = fake.clone().requires_grad_(True)
    loss_G = lossf.compute_loss_G(netD, real, fk)
    loss_G.backward()
    with torch.no_grad():
        feats = netD.freeze_feature_network(torch.nn.functional.interpolate(real, interp, mode="bilinear", align_corners=False))
    torch.save(dict(cfg=dict(S=S, interp=interp, B=B), keys=list(ref_sd.keys()), shapes={k: tuple(v.shape) for k, v in ref_sd.items()},
                    real=real, fake=fake, pred_real=pred_real, loss_D=loss_D.detach(), grad_checks=checks(grads),
                    grad_sample={k: grads[k].flatten()[:6].clone() for k in list(grads)[:6]},
                    feat_checks=checks({k: v for k, v in feats.items()}), feat_shapes={k: tuple(v.shape) for k, v in feats.items()},
                    loss_G=loss_G.detach(), dfake=fk.grad.detach().clone()),
               os.path.join(OUT, out_name))
    print(out_name, "keys", len(ref_sd), "trainable", len(grads), "loss_D", float(loss_D), "loss_G", float(loss_G), "logits", tuple(pred_real.shape),
          "features", {k: tuple(v.shape) for k, v in feats.items()})


if __name__ == "__main__":
    main()
    main(256, "projd_vit256.pt")

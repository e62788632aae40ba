"""GAN objectives of the CUT path on the HIP ops: /root/reference/models/modules/loss.py `GANLoss` (:11-85) for
gan_mode='lsgan' (the train_gan_mode default: MSE against 1 / 0), 'vanilla' (BCE with logits), 'wgangp' (-/+ mean; the reference never
adds its gradient penalty) and 'projected' (the hinge objective :77-84 that
`set_discriminators_info` forces for projected discriminators, base_gan_model.py:544-545), and `DiscriminatorGANLoss` (:249-313)
without APA / D-diffusion augmentation.
lsgan predictions are NHWC logit maps whose channel 0 is valid (PatchGAN output padded to 8 channels); projected predictions are the
concatenated logits [B, N] of the mini-discriminators (every element valid)."""
from __future__ import annotations

import torch.nn as nn

import os

import torch

from .. import ops

BATCH_REAL_FAKE = os.environ.get("JG_D_BATCH_REAL_FAKE", "1") != "0"


class GANLoss(nn.Module):
    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        if gan_mode not in ("lsgan", "vanilla", "wgangp", "projected"):
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)
        self.gan_mode = gan_mode
        self.real_label, self.fake_label = float(target_real_label), float(target_fake_label)

    def __call__(self, prediction, target_is_real, relu=True):
        """loss.py:59-85: lsgan: nn.MSELoss()(prediction, label.expand_as(prediction)); projected: hinge (`relu`) / -mean (generator)."""
        if self.gan_mode == "projected":
            from .projected_d import hinge_loss

            return hinge_loss(prediction, target_is_real, relu)
        if self.gan_mode == "wgangp":       # :72-76: the sign is the label, whatever real_label / fake_label are
            return ops.gan_loss(prediction, "wgangp", 1.0 if target_is_real else 0.0)
        return ops.gan_loss(prediction, self.gan_mode, self.real_label if target_is_real else self.fake_label)


class DiscriminatorGANLoss(nn.Module):
    """loss.py:249-313 (`compute_loss_D` :288-307, `compute_loss_G` :309-313)."""

    def __init__(self, netD, device, train_gan_mode="lsgan", dataaug_D_label_smooth=False, dataaug_APA=False,
                 dataaug_D_diffusion=False):
        super().__init__()
        if dataaug_APA or dataaug_D_diffusion:
            raise NotImplementedError("APA / D-diffusion augmentation are outside the built path")
        self.netD, self.device = netD, device
        self.gan_mode = train_gan_mode
        self.criterionGAN = GANLoss(train_gan_mode, target_real_label=0.9 if dataaug_D_label_smooth else 1.0)
        self.adaptive_pseudo_augmentation_p, self.adjust = 0.0, 0

    def compute_loss_D(self, netD, real, fake, fake_2=None):
        self.real, self.fake = real, fake
        if BATCH_REAL_FAKE and getattr(netD, "per_sample", False) and real.shape == fake.shape and real.dtype == fake.dtype:
            # round 6: a discriminator that is a per-sample function with no state tied to the call (no BatchNorm statistics, no spectral-norm
            # power iteration: the ViT projector with its MLP heads) sees real and fake as ONE batch -- half the launches of the
            # discriminator half, GEMMs of twice the rows; the two losses are taken on the halves of the logits (loss.py:288-307 calls
            # netD twice; same function, the weight gradients of the heads are summed inside one GEMM instead of over two)
            n = real.shape[0]
            pred = netD(torch.cat((self.real, self.fake.detach()), dim=0))
            self.pred_real = pred[:n]
            self.loss_D_real = self.criterionGAN(self.pred_real, True)
            return (self.loss_D_real + self.criterionGAN(pred[n:], False)) * 0.5
        self.pred_real = netD(self.real)
        self.loss_D_real = self.criterionGAN(self.pred_real, True)
        pred_fake = netD(self.fake.detach())
        loss_D_fake = self.criterionGAN(pred_fake, False)
        return (self.loss_D_real + loss_D_fake) * 0.5

    def compute_loss_G(self, netD, real, fake):
        self.real, self.fake = real, fake
        return self.criterionGAN(netD(self.fake), True, relu=False)

    def update(self, niter):
        pass

"""cm_model (consistency model) training step on MI355X: mirror of /root/reference/models/cm_model.py
(`__init__` :115-263, `set_input` :265-351 (inpainting / pix2pix, cond_image_creation="y_t"), `compute_cm_loss`
:353-375 without perceptual terms, `pseudo_huber_loss` :27-43) and models/diffusion_networks.py `define_G`
(:24-139,377-383) for `model_type="cm"`, `G_netG="unet_mha"`.

The step = 2 UNet forwards (student with gradient, teacher without) + 1 backward on the same fused schedule and
kernels as palette_model; the consistency loss and its gradient are one fused kernel (`jg_cm_loss`).
"""
from __future__ import annotations

import torch

from .. import ops
from ..modules.cm_generator import CMGenerator
from ..modules.unet_generator_attn import UNet
from .base_model import BaseModel, NetworkGroup


def define_G_cm(opt):
    """models/diffusion_networks.py:104-139,377-383 restricted to cm + unet_mha."""
    if opt.G_netG != "unet_mha":
        raise NotImplementedError(f"G_netG={opt.G_netG!r}: only unet_mha is built")
    in_channel = opt.model_input_nc
    if (opt.alg_diffusion_cond_embed != "" and opt.alg_diffusion_cond_embed != "y_t") or opt.alg_diffusion_task == "pix2pix":
        in_channel = opt.model_input_nc + opt.model_output_nc
    if "mask" in opt.alg_diffusion_cond_embed:
        raise NotImplementedError("mask conditioning is outside the SURVEY.md 8 hot path")
    model = UNet(
        image_size=opt.data_crop_size, in_channel=in_channel, inner_channel=opt.G_ngf, out_channel=opt.model_output_nc,
        res_blocks=opt.G_unet_mha_res_blocks, attn_res=opt.G_unet_mha_attn_res, num_heads=opt.G_unet_mha_num_heads,
        num_head_channels=opt.G_unet_mha_num_head_channels, tanh=False, dropout=opt.G_dropout,
        n_timestep_train=opt.G_diff_n_timestep_train, n_timestep_test=opt.G_diff_n_timestep_test,
        channel_mults=opt.G_unet_mha_channel_mults, norm=opt.G_unet_mha_norm_layer,
        group_norm_size=opt.G_unet_mha_group_norm_size, efficient=opt.G_unet_mha_vit_efficient,
        cond_embed_dim=opt.alg_diffusion_cond_embed_dim, freq_space=opt.train_feat_wavelet,
    )
    return CMGenerator(cm_model=model, sampling_method="", image_size=opt.data_crop_size, G_ngf=opt.G_ngf, opt=opt)


class CMModel(BaseModel):
    overlap_exchange = True   # one backward per optimizer step: the gradient all-reduce starts inside the backward

    def __init__(self, opt, rank):
        super().__init__(opt, rank)
        self.task = opt.alg_diffusion_task
        if self.task not in ("inpainting", "pix2pix"):
            raise NotImplementedError(f"alg_diffusion_task={self.task!r} is outside the SURVEY.md 8 hot path")
        if opt.alg_diffusion_cond_image_creation != "y_t":
            raise NotImplementedError("only alg_diffusion_cond_image_creation='y_t' is implemented")
        if [p for p in getattr(opt, "alg_cm_perceptual_loss", [""]) if p]:
            raise NotImplementedError("LPIPS / DISTS perceptual terms need pretrained networks (not available offline)")
        self.total_t = opt.alg_cm_num_steps * opt.train_batch_size                       # cm_model.py:129-131
        opt.alg_palette_sampling_method = ""                                              # :193-198
        opt.alg_diffusion_cond_embed = opt.alg_diffusion_cond_image_creation
        opt.alg_diffusion_cond_embed_dim = 32 if opt.alg_diffusion_ddpm_cm_ft else 256
        self.netG_A = define_G_cm(opt)
        if opt.isTrain:
            self.netG_A.current_t = max(self.netG_A.current_t, getattr(opt, "total_iters", 0))   # :200-203
        self.model_names = ["G_A"]
        if opt.isTrain:
            self.optimizer_G = self.make_optimizer(self.netG_A, lr=opt.train_G_lr, betas=(opt.train_beta1, opt.train_beta2),
                                                   weight_decay=opt.train_optim_weight_decay, eps=opt.train_optim_eps)
            self.optimizers.append(self.optimizer_G)
        self.loss_names_G = ["G_tot"]
        self.loss_names = list(self.loss_names_G)
        self.group_G = NetworkGroup(networks_to_optimize=["G_A"], forward_functions=[], backward_functions=["compute_cm_loss"],
                                    loss_names_list=["loss_names_G"], optimizer=["optimizer_G"], loss_backward=["loss_G_tot"],
                                    networks_to_ema=["G_A"])
        self.networks_groups = [self.group_G]
        self.iter_calculator_init()
        self.rng_injection = None  # parity runs: callable(sigmas on the host) -> (noise, timesteps)
        # visuals (cm_model.py:139-185)
        self.gen_visual_names = ["gt_image_", "y_t_", "next_noisy_x_", "current_noisy_x_", "mask_", "output_"]
        for k in range(opt.train_batch_size):
            self.visual_names.append([n + str(k) for n in self.gen_visual_names])
        self.visual_names.append([])
        self.sampling_noises = None   # parity runs: the N(0,1) draw of every sampling sigma

    # cm_model.py:265-351 (4-D inputs)
    def set_input(self, data):
        a = data["A"].to(self.device, non_blocking=True)
        if a.dim() != 4:
            raise NotImplementedError("temporal (5-D) batches are outside the SURVEY.md 8 hot path")
        self.y_t = a
        self.gt_image = data["B"].to(self.device, non_blocking=True)
        self.mask = data["B_label_mask"].to(self.device, non_blocking=True) if self.task == "inpainting" else None
        self.cond_image = self.y_t if self.task == "pix2pix" else None
        self.batch_size = self.y_t.shape[0]
        self.real_A, self.real_B = self.y_t, self.gt_image

    # cm_model.py:353-375
    def compute_cm_loss(self):
        net = self._net("G_A")
        noise = timesteps = None
        if self.rng_injection is not None:
            noise, timesteps = self.rng_injection(self.gt_image.shape[0])
        r = net.forward_nhwc(self.gt_image, self.total_t, self.mask, self.cond_image, noise, timesteps)
        self.next_noisy_x, self.current_noisy_x = r["next_noisy_x"], r["current_noisy_x"]
        self.loss_G_tot = ops.cm_loss(r["F_next"], r["F_cur"], r["next_noisy_x"], r["current_noisy_x"], r["cs_n"], r["co_n"],
                                      r["cs_c"], r["co_c"], self.mask, r["loss_weights"], lam=self.opt.alg_diffusion_lambda_G,
                                      grad_scale=self.loss_scale)

    # cm_model.py:504-657
    SAMPLING_SIGMAS = (80.0, 24.4, 5.84, 0.9, 0.661)

    @torch.no_grad()
    def inference(self, nb_imgs, offset=0):
        netG = self._net("G_A")
        mask = self.mask[:nb_imgs] if self.mask is not None else None
        y_cond = self.cond_image[:nb_imgs] if self.cond_image is not None else None
        if self.task == "pix2pix":      # y_t must have the output's channel count: there is no ground truth to start from
            shp = list(y_cond.shape)
            shp[1] = netG.cm_model.out_channel
            y_t = torch.zeros(shp, device=y_cond.device, dtype=y_cond.dtype)
        else:
            y_t = self.y_t[:nb_imgs]
        self.output = netG.restoration(y_t, y_cond, self.SAMPLING_SIGMAS, mask, noises=self.sampling_noises)
        self.fake_B = self.output
        self.visuals = self.output
        self._publish_visuals(nb_imgs, offset)

    def compute_visuals(self, nb_imgs):
        super().compute_visuals(nb_imgs)
        self.inference(nb_imgs)

    def get_current_visuals(self, nb_imgs, phase="train", test_name=""):
        old = self.visual_names.copy()
        if phase == "test":             # the noisy columns are hidden in test mode
            self.visual_names = [[x for x in grp if "noisy" not in x] for grp in self.visual_names]
        out = super().get_current_visuals(nb_imgs, phase, test_name)
        self.visual_names = old
        return out

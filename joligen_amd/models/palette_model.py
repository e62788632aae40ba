"""palette_model (DDPM) training step on MI355X: mirror of /root/reference/models/palette_model.py
(`__init__` :116-285, `set_input` :287-366 (inpainting / pix2pix, cond_image_creation="y_t"),
`compute_palette_loss` :558-620) and models/diffusion_networks.py `define_G` (:24-139,361-376)
for `G_netG="unet_mha"`.

Same public surface as the reference model class: `set_input(data)`,
`optimize_parameters()`, `get_current_losses()`, `save_networks()`, `netG_A`, `model_names`...
Inference / sampling (`inference` :622-887) is the first "next" row of SURVEY.md 8(f).
"""
from __future__ import annotations

import torch

from .. import ops
from ..modules.diffusion_generator import DiffusionGenerator, PaletteDenoiseFn
from ..modules.unet_generator_attn import UNet
from .base_model import BaseModel, NetworkGroup


def define_G(model_type, model_input_nc, model_output_nc, G_netG, data_crop_size, G_diff_n_timestep_train,
             G_diff_n_timestep_test, G_dropout, G_ngf, G_unet_mha_num_heads, G_unet_mha_num_head_channels,
             G_unet_mha_res_blocks, G_unet_mha_channel_mults, G_unet_mha_attn_res, G_unet_mha_norm_layer,
             G_unet_mha_group_norm_size, G_unet_mha_vit_efficient, alg_palette_sampling_method, alg_diffusion_cond_embed,
             alg_diffusion_cond_embed_dim, alg_diffusion_ref_embed_net="clip", model_prior_321_backwardcompatibility=False,
             f_s_semantic_nclasses=-1, train_feat_wavelet=False, **unused_options):
    """models/diffusion_networks.py:24-139,361-376 restricted to palette + unet_mha."""
    if model_type != "palette":
        raise NotImplementedError(f"define_G(model_type={model_type!r}) not implemented yet")
    if G_netG != "unet_mha":
        raise NotImplementedError(f"G_netG={G_netG!r}: only unet_mha is on the SURVEY.md 8 DDPM path")
    in_channel = model_input_nc + model_output_nc
    if "mask" in alg_diffusion_cond_embed:
        in_channel += alg_diffusion_cond_embed_dim
    model = UNet(
        image_size=data_crop_size, in_channel=in_channel, inner_channel=G_ngf, out_channel=model_output_nc,
        res_blocks=G_unet_mha_res_blocks, attn_res=G_unet_mha_attn_res, num_heads=G_unet_mha_num_heads,
        num_head_channels=G_unet_mha_num_head_channels, tanh=False, dropout=G_dropout,
        n_timestep_train=G_diff_n_timestep_train, n_timestep_test=G_diff_n_timestep_test,
        channel_mults=G_unet_mha_channel_mults, norm=G_unet_mha_norm_layer, group_norm_size=G_unet_mha_group_norm_size,
        efficient=G_unet_mha_vit_efficient, cond_embed_dim=alg_diffusion_cond_embed_dim, freq_space=train_feat_wavelet,
    )
    denoise_fn = PaletteDenoiseFn(model=model, cond_embed_dim=alg_diffusion_cond_embed_dim,
                                  ref_embed_net=alg_diffusion_ref_embed_net, conditioning=alg_diffusion_cond_embed,
                                  nclasses=f_s_semantic_nclasses)
    return DiffusionGenerator(denoise_fn=denoise_fn, sampling_method=alg_palette_sampling_method,
                              image_size=data_crop_size, G_ngf=G_ngf,
                              loading_backward_compatibility=model_prior_321_backwardcompatibility)


class PaletteModel(BaseModel):
    overlap_exchange = True   # one backward per optimizer step: the gradient all-reduce starts inside the backward

    def __init__(self, opt, rank):
        super().__init__(opt, rank)
        self.task = opt.alg_diffusion_task
        if self.task not in ("inpainting", "pix2pix"):
            raise NotImplementedError(f"alg_diffusion_task={self.task!r} is outside the SURVEY.md 8 hot path")
        if opt.alg_diffusion_cond_image_creation != "y_t":
            raise NotImplementedError("only alg_diffusion_cond_image_creation='y_t' is implemented")
        if opt.alg_palette_loss not in ("MSE", "L1", "multiscale_MSE", "multiscale_L1"):
            raise NotImplementedError(f"alg_palette_loss={opt.alg_palette_loss!r}")
        if "ref" in opt.alg_diffusion_cond_embed:
            raise NotImplementedError("reference-image conditioning needs CLIP / ImageBind encoders (not built)")
        if opt.isTrain and opt.alg_diffusion_dropout_prob > 0 and not getattr(opt, "_jg_dropout_class_added", False):
            # PaletteModel.after_parse (palette_model.py:102-107): one more class, the unconditioned one
            opt.f_s_semantic_nclasses += 1
            opt.cls_semantic_nclasses += 1
            opt._jg_dropout_class_added = True
        if getattr(opt, "alg_diffusion_generate_per_class", False):
            raise NotImplementedError("alg_diffusion_generate_per_class visuals are not built")
        self.num_classes = max(opt.f_s_semantic_nclasses, opt.cls_semantic_nclasses)      # :149-151
        self.drop_injection = None      # parity runs: callable(B) -> the uniform draws of the conditioning dropout
        if opt.G_nblocks == 9 and "resnet" not in opt.G_netG:
            opt.G_nblocks = 2  # palette_model.py:199-203

        self.netG_A = define_G(**vars(opt))
        self.model_names = ["G_A"]
        if opt.isTrain:
            self.optimizer_G = self.make_optimizer(self.netG_A, lr=opt.train_G_lr, betas=(opt.train_beta1, opt.train_beta2),
                                                   weight_decay=opt.train_optim_weight_decay, eps=opt.train_optim_eps)
            self.optimizers.append(self.optimizer_G)
        self.loss_names_G = ["G_tot"]
        if "multiscale" in opt.alg_palette_loss:      # palette_model.py:231-241
            import math

            S = opt.data_crop_size
            self.loss_names_G += ["G_%d" % (2 ** k) for k in range(5, math.floor(math.log2(S)) + 1)] + ["G_%d" % S]
        self.loss_names = list(self.loss_names_G)
        self.group_G = NetworkGroup(networks_to_optimize=["G_A"], forward_functions=[],
                                    backward_functions=["compute_palette_loss"], loss_names_list=["loss_names_G"],
                                    optimizer=["optimizer_G"], loss_backward=["loss_G_tot"], networks_to_ema=["G_A"])
        self.networks_groups = [self.group_G]
        self.iter_calculator_init()
        self.rng_injection = None  # parity runs: callable(batch_size) -> (t, u, noise) drawn on the host
        # visuals (palette_model.py:158-198) and the sampler settings of `inference` (:281-285)
        self.gen_visual_names = ["gt_image_", "cond_image_"] + (["y_t_", "mask_"] if self.task == "inpainting" else []) + ["output_"]
        if opt.isTrain:
            for k in range(min(opt.train_batch_size, getattr(opt, "output_num_images", 20))):
                self.visual_names.append([n + str(k) for n in self.gen_visual_names])
        self.visual_names.append([])
        self.sample_num = 2
        self.ddim_num_steps = getattr(opt, "alg_palette_ddim_num_steps", 10)
        self.ddim_eta = getattr(opt, "alg_palette_ddim_eta", 0.5)
        self.sampling_noises = None   # parity runs: per-step N(0,1) draws of the ancestral sampler

    # palette_model.py:287-366 (4-D inputs, no SAM masks, no reference image)
    def set_input(self, data):
        a = data["A"].to(self.device, non_blocking=True)
        if a.dim() != 4:
            raise NotImplementedError("temporal (5-D) batches are outside the SURVEY.md 8 hot path")
        if self.task == "inpainting":
            self.y_t = a
            self.gt_image = data["B"].to(self.device, non_blocking=True)
            self.mask = data["B_label_mask"].to(self.device, non_blocking=True)
        else:  # pix2pix
            self.y_t = a
            self.gt_image = data["B"].to(self.device, non_blocking=True)
            self.mask = None
        self.cls = data["B_label_cls"].to(self.device).long() if "B_label_cls" in data else None      # :368-371
        self.cond_image = self.y_t
        self.batch_size = self.cond_image.shape[0]
        self.real_A = self.cond_image
        self.real_B = self.gt_image

    # palette_model.py:558-620
    def compute_palette_loss(self):
        y_0, y_cond, mask = self.gt_image, self.cond_image, self.mask
        t = u = noise = None
        if self.rng_injection is not None:
            t, u, noise = self.rng_injection(y_0.shape[0])
            noise = noise.to(self.device)
        cls = self.cls
        if self.opt.alg_diffusion_dropout_prob > 0.0:
            # :565-584: the conditioning of a random subset of the batch is replaced by the highest ("unconditioned") class; the draw
            # precedes the generator's (t, u, noise) draws on the same RNG stream
            r = self.drop_injection(y_0.shape[0]).to(self.device) if self.drop_injection is not None else torch.rand(y_0.shape[0], device=self.device)
            drop = r < self.opt.alg_diffusion_dropout_prob
            if cls is not None:
                cls = torch.where(drop, torch.full_like(cls, self.num_classes - 1), cls)
            if mask is not None:     # :573-579: every mask pixel of a dropped sample becomes the highest class too (clamped to 1 downstream)
                mask = torch.where(drop.view(-1, 1, 1, 1).expand(mask.shape), torch.full_like(mask, self.num_classes - 1), mask)
        net = self._net("G_A")
        noise, noise_hat, min_snr_w, _ = net.forward_nhwc(y_0, y_cond, mask, noise, t, u, cls=cls)
        w = min_snr_w if self.opt.alg_palette_minsnr else None
        loss, levels = ops.ddpm_loss(noise_hat, noise.float(), mask, w, lam=self.opt.alg_diffusion_lambda_G,
                                     grad_scale=self.loss_scale, lossname=self.opt.alg_palette_loss)
        lam = self.opt.alg_diffusion_lambda_G
        for res, val in levels.items():          # palette_model.py:610-616: the per-resolution terms are logged before lambda_G
            setattr(self, "loss_G_" + res, val / lam if lam not in (0, 1) else val)
        self.loss_G_tot = loss

    # palette_model.py:622-887 (inpainting / pix2pix; no per-class, reference-image or video branches)
    @torch.no_grad()
    def inference(self, nb_imgs, offset=0):
        netG = self._net("G_A")
        cls = self.cls[:nb_imgs] if self.cls is not None else None        # palette_model.py:739-760 (no per-class generation)
        if self.task == "inpainting":
            self.output, self.visuals = netG.restoration(y_cond=self.cond_image[:nb_imgs], y_t=self.y_t[:nb_imgs], y_0=self.gt_image[:nb_imgs],
                                                         mask=self.mask[:nb_imgs], sample_num=self.sample_num, cls=cls,
                                                         ddim_num_steps=self.ddim_num_steps, ddim_eta=self.ddim_eta, noises=self.sampling_noises)
        else:
            self.output, self.visuals = netG.restoration(y_cond=self.cond_image[:nb_imgs], sample_num=self.sample_num, cls=cls,
                                                         noises=self.sampling_noises)
        self.fake_B = self.output
        self._publish_visuals(nb_imgs, offset)

    def compute_visuals(self, nb_imgs):
        super().compute_visuals(nb_imgs)
        self.inference(nb_imgs)

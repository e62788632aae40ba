"""cut_model (contrastive unpaired translation) training step on MI355X: mirror of /root/reference/models/cut_model.py
(`__init__` :216-503, `data_dependent_initialize` :505-548, `forward_cut` :611-688, `compute_G_loss_cut` :708-837,
`calculate_feats` :848-887, `calculate_NCE_loss` :889-909) and models/base_gan_model.py (`compute_D_loss(_generic)`
:341-419, `compute_G_loss_GAN(_generic)` :421-503) for the default configuration: G_netG='resnet', D_netDs=['basic'],
alg_cut_netF='mlp_sample', alg_cut_nce_loss in {'monce', 'patchnce'}, nce_idt, lsgan, no semantic / multimodal / context /
temporal / augmentation branches.

Per iteration (reference order): group G = {G_A, F}: fake = G(cat(real_A, real_B)); loss_G_tot = lambda_GAN * lsgan(D(fake_B), 1)
+ (NCE(real_A, fake_B) + NCE(real_B, idt_B)) / 2; backward; Adam step on G and on F (two fused launches).  Group D: fake from the
image pool, loss_D_tot = (lsgan(D(real_B), 1) + lsgan(D(fake), 0)) / 2; backward; Adam step on D.
Images are converted ONCE to NHWC 16-bit (3 -> 8 channels); every conv / norm / loss runs on the HIP kernels."""
from __future__ import annotations

import torch

from .._autograd import JGFunction
from .. import ops
from ..modules.NCE.patchnce import MoNCELoss, PatchNCELoss
from ..modules.cut_networks import PatchSampleF
from ..modules.discriminators import NLayerDiscriminator
from ..modules.loss import DiscriminatorGANLoss
from ..modules.resnet_generator import ResnetGenerator
from ..util.image_pool import ImagePool
from .base_model import BaseModel, NetworkGroup

CUT_DEFAULTS = dict(
    alg_cut_lambda_NCE=1.0, alg_cut_lambda_SRC=0.0, alg_cut_nce_idt=True, alg_cut_nce_layers="0,4,8,12,16",
    alg_cut_nce_includes_all_negatives_from_minibatch=False, alg_cut_nce_loss="monce", alg_cut_netF="mlp_sample",
    alg_cut_netF_nc=256, alg_cut_nce_T=0.07, alg_cut_num_patches=256, alg_cut_flip_equivariance=False, alg_cut_MSE_idt=False,
    alg_cut_supervised_loss=[""], alg_gan_lambda=1.0, train_gan_mode="lsgan", D_netDs=["basic"], D_ndf=64, D_n_layers=3,
    D_dropout=False, D_spectral=False, dataaug_D_label_smooth=False, dataaug_D_noise=0.0, dataaug_APA=False,
    dataaug_D_diffusion=False, train_semantic_mask=False, train_semantic_cls=False, train_mask_out_mask=False,
)


class _ScaleGradFn(JGFunction):
    """identity on the loss value; multiplies the gradient by the static fp16 loss scale (1 for bf16)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


class CUTModel(BaseModel):
    def __init__(self, opt, rank):
        for k, v in CUT_DEFAULTS.items():
            if not hasattr(opt, k):
                setattr(opt, k, v)
        super().__init__(opt, rank)
        if self.act_dtype == torch.float16 and not float(getattr(opt, "jg_loss_scale", 0.0) or 0.0):
            # the 1/T = 14x of the contrastive logits makes these gradients ~2 orders larger than the diffusion path's:
            # 65536 overflows fp16 activation gradients, 1024 keeps both ends of the range
            self.loss_scale = 1024.0
        if opt.G_netG not in ("resnet", "resnet_9blocks", "resnet_6blocks", "segformer_attn_conv", "resnet_attn", "mobile_resnet_attn"):
            raise NotImplementedError(f"G_netG={opt.G_netG!r}: the CUT path is built for the resnet, resnet_attn, mobile_resnet_attn and "
                                      "segformer_attn_conv generators")
        if "segformer" in opt.G_netG:           # cut_model.py:205-210: enforced by the reference
            opt.alg_cut_nce_layers, opt.alg_cut_nce_T = "0,1,2,3", 0.2
        bad = [d for d in opt.D_netDs if d not in ("basic", "projected_d")]
        if bad or not opt.D_netDs:
            raise NotImplementedError(f"D_netDs={opt.D_netDs!r}: 'basic' (PatchGAN) and 'projected_d' are built ('vision_aided' and the "
                                      "depth / mask / sam / temporal discriminators need pretrained networks)")
        if opt.alg_cut_netF != "mlp_sample":
            raise NotImplementedError(f"alg_cut_netF={opt.alg_cut_netF!r}")
        if opt.alg_cut_nce_loss not in ("monce", "patchnce"):
            raise NotImplementedError(f"alg_cut_nce_loss={opt.alg_cut_nce_loss!r}")
        for flag in ("model_multimodal", "alg_cut_flip_equivariance", "alg_cut_MSE_idt", "train_semantic_mask", "train_semantic_cls",
                     "train_mask_out_mask", "dataaug_APA", "dataaug_D_diffusion"):
            if getattr(opt, flag, False):
                raise NotImplementedError(f"{flag} is outside the SURVEY.md 8 hot path")
        if opt.alg_cut_lambda_SRC > 0 or [s for s in opt.alg_cut_supervised_loss if s] or opt.dataaug_D_noise > 0:
            raise NotImplementedError("SRC / supervised / noisy-D terms are outside the built path")
        # options that select ANOTHER network than the one built here must not be dropped silently: the generator /
        # PatchGAN below are the InstanceNorm, no-dropout, no-spectral-norm variants of gan_networks.define_G / define_D
        for name, built in (("G_dropout", False), ("D_dropout", False), ("D_spectral", False), ("G_spectral", False),
                            ("G_norm", "instance"), ("D_norm", "instance")):
            val = getattr(opt, name, built)
            if val != built and not (name.endswith("_norm") and "segformer" in opt.G_netG and name == "G_norm"):
                raise NotImplementedError(f"{name}={val!r}: only {built!r} is built for the CUT networks")
        self.nce_layers = [int(i) for i in str(opt.alg_cut_nce_layers).split(",")]
        if "segformer" in opt.G_netG:            # gan_networks.py:177-187
            from ..modules.segformer import SegformerGenerator_attn

            self.netG_A = SegformerGenerator_attn(getattr(opt, "jg_dir", ""), getattr(opt, "G_config_segformer", ""), opt.model_input_nc,
                                                  img_size=opt.data_crop_size, nb_mask_attn=getattr(opt, "G_attn_nb_mask_attn", 10),
                                                  nb_mask_input=getattr(opt, "G_attn_nb_mask_input", 1), final_conv=True,
                                                  padding_type=opt.G_padding_type)
        elif "resnet_attn" in opt.G_netG:        # gan_networks.py:150-176
            from ..modules.resnet_attn_generator import ResnetGenerator_attn

            if getattr(opt, "train_feat_wavelet", False):
                raise NotImplementedError("train_feat_wavelet (wavelet feature space) is outside the built path")
            self.netG_A = ResnetGenerator_attn(opt.model_input_nc, opt.model_output_nc, getattr(opt, "G_attn_nb_mask_attn", 10),
                                               getattr(opt, "G_attn_nb_mask_input", 1), opt.G_ngf, n_blocks=opt.G_nblocks,
                                               padding_type=opt.G_padding_type, mobile=opt.G_netG.startswith("mobile"),
                                               twice_resnet_blocks=getattr(opt, "G_backward_compatibility_twice_resnet_blocks", False))
        else:
            self.netG_A = ResnetGenerator(opt.model_input_nc, opt.model_output_nc, opt.G_ngf, n_blocks=opt.G_nblocks,
                                          padding_type=opt.G_padding_type)
        self.model_names = ["G_A"]
        if opt.isTrain:
            self.netF = PatchSampleF(use_mlp=True, init_type=opt.model_init_type, init_gain=opt.model_init_gain, nc=opt.alg_cut_netF_nc)
            self.netF.set_device(self.device)
            # gan_networks.define_D (:330-446): one network per entry of D_netDs, named D_B_<entry>
            self.discriminators_names = []
            for d in opt.D_netDs:
                if d == "basic":
                    net = NLayerDiscriminator(opt.model_output_nc, opt.D_ndf, n_layers=opt.D_n_layers)
                else:
                    from ..modules.projected_d import ProjectedDiscriminator

                    # jg_projd_backbone: "lite0" (tf_efficientnet_lite0, the reference's feature network) | "standin" (tests);
                    # jg_projd_pretrained: path of a timm tf_efficientnet_lite0 state_dict (the weights cannot be downloaded here)
                    net = ProjectedDiscriminator(getattr(opt, "D_proj_network_type", "efficientnet"), interp=getattr(opt, "D_proj_interp", -1),
                                                 img_size=opt.data_crop_size, backbone=getattr(opt, "jg_projd_backbone", "lite0"),
                                                 pretrained_path=getattr(opt, "jg_projd_pretrained", ""))
                setattr(self, "netD_B_" + d, net)
                self.discriminators_names.append("D_B_" + d)
            self.model_names += ["F"] + self.discriminators_names
            # base_model.py:115-118; forward_GAN (base_gan_model.py:170-173) also pushes the real images through pools on every
            # iteration: they feed only the metrics, but they consume host random draws BEFORE the fake pool does
            self.real_A_pool, self.real_B_pool = ImagePool(opt.train_pool_size), ImagePool(opt.train_pool_size)
            self.fake_B_pool = ImagePool(opt.train_pool_size)
            crit = MoNCELoss if opt.alg_cut_nce_loss == "monce" else PatchNCELoss
            self.criterionNCE = [crit(opt) for _ in self.nce_layers]
            kw = dict(lr=opt.train_G_lr, betas=(opt.train_beta1, opt.train_beta2), weight_decay=opt.train_optim_weight_decay,
                      eps=opt.train_optim_eps)
            self.optimizer_G = self.make_optimizer(self.netG_A, **kw)
            kw["lr"] = opt.train_D_lr
            # the reference chains every discriminator's parameters into ONE Adam (cut_model.py:378-395); one fused optimizer per
            # discriminator arena with the same hyper-parameters is the same update
            optD = []
            for dn in self.discriminators_names:
                o = self.make_optimizer(getattr(self, "net" + dn), **kw)
                setattr(self, "optimizer_" + dn, o)
                optD.append("optimizer_" + dn)
                self.optimizers.append(o)
                # base_gan_model.set_discriminators_info (:538-640): projected discriminators always train with the hinge objective
                mode = "projected" if "projected" in dn else opt.train_gan_mode
                calc = DiscriminatorGANLoss(getattr(self, "net" + dn), self.device, mode, opt.dataaug_D_label_smooth)
                setattr(self, dn + "_loss_calculator", calc)
                self.objects_to_update.append(calc)
            self.optimizer_D = getattr(self, optD[0])
            self.optimizers.append(self.optimizer_G)
            self.group_G = NetworkGroup(networks_to_optimize=["G_A", "F"], forward_functions=["forward"],
                                        backward_functions=["compute_G_loss"], loss_names_list=["loss_names_G"],
                                        optimizer=["optimizer_G", "optimizer_F"], loss_backward=["loss_G_tot"], networks_to_ema=["G_A"])
            self.group_D = NetworkGroup(networks_to_optimize=list(self.discriminators_names), forward_functions=None,
                                        backward_functions=["compute_D_loss"], loss_names_list=["loss_names_D"], optimizer=optD,
                                        loss_backward=["loss_D_tot"])
            self.networks_groups = [self.group_G, self.group_D]
            self.loss_names_G = (["G_tot", "G_NCE", "G_NCE_Y"] if opt.alg_cut_nce_idt else ["G_tot", "G_NCE"]) + \
                ["G_GAN_" + dn for dn in self.discriminators_names]
            self.loss_names_D = ["D_tot"] + ["D_GAN_" + dn for dn in self.discriminators_names]
            self.loss_names = self.loss_names_G + self.loss_names_D
            self.loss_functions_G = ["compute_G_loss_GAN", "compute_G_loss_cut"]
            self.iter_calculator_init()
        else:
            self.netG_A.jg_finalize(self.device, self.act_dtype)
        self.patch_ids_injection = None   # parity runs: callable(call_index, feat_shapes) -> list of id tensors

    # ---- inputs ---------------------------------------------------------------------------------------------------
    def set_input(self, data):
        self.real_A_nchw = data["A"].to(self.device, non_blocking=True)
        self.real_B_nchw = data["B"].to(self.device, non_blocking=True)
        self.real_A = ops.to_nhwc(self.real_A_nchw, self.act_dtype)
        self.real_B = ops.to_nhwc(self.real_B_nchw, self.act_dtype)
        self.batch_size = self.real_A.shape[0]

    def get_current_batch_size(self):
        return self.batch_size

    def data_dependent_initialize(self, data):
        """cut_model.py:505-548: the MLP widths of netF come from the tapped feature widths; optimizer_F is created here."""
        self.set_input(data)
        if self.opt.isTrain:
            self.feat_channels = self.netG_A.feat_channels(self.nce_layers)
            self.netF.data_dependent_initialize(None, self.feat_channels)
            self.optimizer_F = self.make_optimizer(self.netF, lr=self.opt.train_G_lr, betas=(self.opt.train_beta1, self.opt.train_beta2),
                                                   weight_decay=self.opt.train_optim_weight_decay, eps=self.opt.train_optim_eps)
            self.optimizers.append(self.optimizer_F)
        for o in self.optimizers:
            o.zero_grad()

    # ---- forward (cut_model.py:611-688) ---------------------------------------------------------------------------
    def set_pool_rng(self, rng):
        """parity runs: one host RNG (uniform / randint) shared by the three pools, like the reference's `random` module."""
        for p in (self.real_A_pool, self.real_B_pool, self.fake_B_pool):
            p.rng = rng

    def forward(self):
        B = self.batch_size
        if self.opt.isTrain:
            self.real_A_pool.query(self.real_A)
            self.real_B_pool.query(self.real_B)
        self.real = torch.cat((self.real_A, self.real_B), dim=0) if self.opt.alg_cut_nce_idt else self.real_A
        self.fake = self._net("G_A")(self.real)
        self.fake_B = self.fake[:B]
        if self.opt.alg_cut_nce_idt:
            self.idt_B = self.fake[B:]
        self._feat_calls = 0

    # ---- generator losses ---------------------------------------------------------------------------------------------
    def compute_G_loss(self):
        self.loss_G_tot = 0
        for f in self.loss_functions_G:
            getattr(self, f)()
        self.loss_G_tot = _ScaleGradFn.apply(self.loss_G_tot, self.loss_scale)

    def compute_G_loss_GAN(self):
        """base_gan_model.py:421-503: lambda_GAN * compute_loss_G of every discriminator on domain B."""
        for dn in self.discriminators_names:
            lossf = getattr(self, dn + "_loss_calculator")
            val = self.opt.alg_gan_lambda * lossf.compute_loss_G(self._net(dn), self.real_B, self.fake_B)
            setattr(self, "loss_G_GAN_" + dn, val)
            self.loss_G_tot = self.loss_G_tot + val

    def compute_G_loss_cut(self):
        """cut_model.py:708-837 (NCE + identity NCE)."""
        fq, fk = self.calculate_feats(self.real_A, self.fake_B)
        self.loss_G_NCE = self.calculate_NCE_loss(fq, fk) if self.opt.alg_cut_lambda_NCE > 0.0 else 0.0
        if self.opt.alg_cut_nce_idt and self.opt.alg_cut_lambda_NCE > 0.0:
            fq, fk = self.calculate_feats(self.real_B, self.idt_B)
            self.loss_G_NCE_Y = self.calculate_NCE_loss(fq, fk)
            loss_NCE_both = (self.loss_G_NCE + self.loss_G_NCE_Y) * 0.5
        else:
            loss_NCE_both = self.loss_G_NCE
        self.loss_G_tot = self.loss_G_tot + loss_NCE_both

    def calculate_feats(self, src, tgt):
        """:848-887: q from the translated image, k from the source, SAME patch ids (drawn on the k pass)."""
        net = self._net("G_A")
        feat_q = net.get_feats(tgt, self.nce_layers)
        feat_k = net.get_feats(src, self.nce_layers)
        ids = None
        if self.patch_ids_injection is not None:
            ids = self.patch_ids_injection(self._feat_calls, [tuple(f.shape) for f in feat_k])
        self._feat_calls += 1
        netF = self._net("F")
        netF.arena.ensure_fresh()
        feat_k_pool, sample_ids = netF(feat_k, self.opt.alg_cut_num_patches, ids, self.feat_channels)
        feat_q_pool, _ = netF(feat_q, self.opt.alg_cut_num_patches, sample_ids, self.feat_channels)
        return feat_q_pool, feat_k_pool

    def calculate_NCE_loss(self, feat_q_pool, feat_k_pool):
        """:889-909."""
        total = 0.0
        for f_q, f_k, crit in zip(feat_q_pool, feat_k_pool, self.criterionNCE):
            loss = crit(feat_q=f_q, feat_k=f_k, current_batch=self.get_current_batch_size()) * self.opt.alg_cut_lambda_NCE
            total = total + loss.mean()
        return total / len(self.nce_layers)

    # ---- discriminator loss (base_gan_model.py:341-419) ------------------------------------------------------------------
    def compute_D_loss(self):
        """base_gan_model.py:341-419: every discriminator draws ITS OWN fake batch from the history pool (compute_D_loss_generic)."""
        tot = 0
        for dn in self.discriminators_names:
            fake = self.fake_B_pool.query(self.fake_B)
            val = getattr(self, dn + "_loss_calculator").compute_loss_D(self._net(dn), self.real_B, fake, None)
            setattr(self, "loss_D_GAN_" + dn, val)
            tot = tot + val
        self.loss_D_tot = _ScaleGradFn.apply(tot, self.loss_scale)

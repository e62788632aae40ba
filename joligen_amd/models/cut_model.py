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

import os

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


FORK_GAN_DEFAULT = True
# round 6: the history-pool draws of the early-D drivers run on the discriminator stream (JG_POOL_SIDE=0: on the main stream, as in round 5)
POOL_ON_SIDE = os.environ.get("JG_POOL_SIDE", "1") != "0"


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
        if self.opt.isTrain:
            self.real_A_pool.store(self.real_A)
            self.real_B_pool.store(self.real_B)
        self._forward_core()

    def _reuse_feats(self):
        """`jg_nce_reuse_feats` (round 6, default on): the key-side features of both contrastive terms -- `netG.get_feats` of the source image
        (NCE term) and of the target image (identity term), cut_model.py:848-887 -- are the encoder activations the generator's forward has
        just computed on cat(real_A, real_B): for an encoder that is a deterministic function of its input in training mode (the ResNet
        families: InstanceNorm, no dropout) a second pass recomputes the same values from the same weights, and its backward adds the same
        gradients to them.  The forward hands the tapped activations out (`forward_with_feats`) and the encoder pass of the loss runs over the
        2 B translated / identity images only.  NOT for the SegFormer encoder: its DropPath masks are drawn per pass in the reference."""
        net = self._net("G_A")
        return (getattr(self.opt, "jg_nce_reuse_feats", True) and os.environ.get("JG_NCE_REUSE_FEATS", "1") != "0" and self.opt.isTrain
                and getattr(net, "deterministic_encoder", False) and net.training and self._batched_nce())

    def _forward_core(self):
        B = self.batch_size
        self.real = torch.cat((self.real_A, self.real_B), dim=0) if self.opt.alg_cut_nce_idt else self.real_A
        self._real_feats = None
        if self._reuse_feats():
            self.fake, self._real_feats = self._net("G_A").forward_with_feats(self.real, self.nce_layers)
        else:
            self.fake = self._net("G_A")(self.real)
        self.fake_B = self.fake[:B]
        if self.opt.alg_cut_nce_idt:
            self.idt_B = self.fake[B:]
        self._feat_calls = 0

    # ---- generator losses ---------------------------------------------------------------------------------------------
    def _fork_gan(self):
        """`JG_FORK_GAN` / option `jg_fork_gan` (round 6): the two branches of the generator loss hang off `fake_B` independently -- the GAN terms
        (every discriminator's forward on the translated image, ViT projector included) and the contrastive terms (encoder passes, PatchSampleF,
        NCE) -- and each is a string of 5 - 30 us launches that fills a fraction of the chip.  The GAN branch is enqueued on a forked stream;
        autograd runs a node's backward on the stream of its forward, so the two branches overlap in the backward as well, and both forks are
        captured into the generator graphs (a fork that joins before the capture ends is a legal capture)."""
        want = os.environ.get("JG_FORK_GAN", "")
        return ((getattr(self.opt, "jg_fork_gan", FORK_GAN_DEFAULT) or want == "1") and want != "0" and self.device.type == "cuda"
                and not ops.TORCH_OPS_BOUNDARY and ops.KERNEL_TIMING is None
                and "compute_G_loss_GAN" in self.loss_functions_G and len(self.loss_functions_G) > 1)

    def compute_G_loss(self):
        self.loss_G_tot = 0
        if self._fork_gan():
            main = torch.cuda.current_stream(self.device)
            side = self.__dict__.get("_gan_stream")
            if side is None:
                side = self._gan_stream = torch.cuda.Stream(device=self.device)
            side.wait_stream(main)
            for t in (self.fake_B, self.real_B):
                t.record_stream(side)
            with torch.cuda.stream(side):
                self.compute_G_loss_GAN()
                gan_tot, self.loss_G_tot = self.loss_G_tot, 0
            for f in self.loss_functions_G:
                if f != "compute_G_loss_GAN":
                    getattr(self, f)()
            main.wait_stream(side)
            gan_tot.record_stream(main)
            for dn in self.discriminators_names:
                getattr(self, "loss_G_GAN_" + dn).record_stream(main)
            self.loss_G_tot = self.loss_G_tot + gan_tot
        else:
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

    def _batched_nce(self):
        """one encoder pass / one PatchSampleF pass / one NCE launch set for BOTH contrastive terms (`jg_batched_nce`, default on): needs the
        identity term, an un-injected random source (parity runs inject per-call draws in the reference's order) and the per-image negatives"""
        o = self.opt
        net = self._net("G_A")
        return (getattr(o, "jg_batched_nce", True) and os.environ.get("JG_BATCHED_NCE", "1") != "0" and o.alg_cut_nce_idt and o.alg_cut_lambda_NCE > 0.0
                and self.patch_ids_injection is None and getattr(getattr(net, "rand", None), "source", None) is None
                and not o.alg_cut_nce_includes_all_negatives_from_minibatch and not ops.TORCH_OPS_BOUNDARY)

    def compute_G_loss_cut_batched(self):
        """cut_model.py:708-909 with the four encoder passes (translated / source image of the NCE term, identity / target image of the identity
        term), the four PatchSampleF passes and the 2 x L contrastive problems each run ONCE on the concatenated batch (round 5).  The reference
        calls `netG.get_feats` four times on B images and `netF` four times; the encoder is per-sample (LayerNorm / InstanceNorm, DropPath draws
        per sample) and netF's MLPs are per-row, so one pass over 4 B images computes the same features -- with a quarter of the launches, on
        problems four times the size.  Patch ids: one draw per (term, layer) in the reference's order (the source-image pass draws, the translated
        image reuses them, cut_networks.py:57-60).  The L problems of a term that share a patch count, and the two terms, are one batched
        PatchNCE / MoNCE call (images are independent problems in every kernel: the 50 Sinkhorn iterations of 2 L x B images run side by side
        instead of in 2 L launches of B)."""
        o = self.opt
        B = self.batch_size
        net, netF = self._net("G_A"), self._net("F")
        # images in the order [translated | identity | source | target] = cat(G's output, G's input): no slice of either is needed, and the
        # rows of every layer come out as [q of term 0 | q of term 1 | k of term 0 | k of term 1]
        reuse = self.__dict__.get("_real_feats")
        if reuse is not None:         # `jg_nce_reuse_feats`: the source / target features are the forward's own activations
            feats = net.get_feats(self.fake, self.nce_layers)
            feats_k, self._real_feats = reuse, None
        else:
            feats = net.get_feats(torch.cat((self.fake, self.real), dim=0), self.nce_layers)
            feats_k = None
        self._feat_calls += 2
        netF.arena.ensure_fresh()
        P = o.alg_cut_num_patches
        rows, counts = [], []
        ids_all = [[netF.draw_ids(f, P) for f in feats] for _ in range(2)]         # term 0: NCE, term 1: identity NCE (reference draw order)
        for li, f in enumerate(feats):
            C = self.feat_channels[li]
            ids = torch.stack((ids_all[0][li], ids_all[1][li]))                      # [2, P_l]: images b use set (b // B) % 2
            if feats_k is not None:
                g = torch.cat((ops.gather_patches(f, ids, C, per=B), ops.gather_patches(feats_k[li], ids, C, per=B)), dim=0)
            else:
                g = ops.gather_patches(f, ids, C, per=B)
            rows.append(netF.embed(g, li).split(2 * B * ids.shape[1]))     # (q rows, k rows): one cat in backward
            counts.append(ids.shape[1])
        T, monce = o.alg_cut_nce_T, o.alg_cut_nce_loss == "monce"
        tot = [0.0, 0.0]

        def one_set(Pl):
            ls = [i for i, c in enumerate(counts) if c == Pl]
            n = B * Pl
            q = torch.cat([rows[i][0] for i in ls], dim=0)                           # [layer][term][B * P_l] problems
            k = torch.cat([rows[i][1] for i in ls], dim=0)
            loss = ops.patch_nce_loss(q, k, 2 * len(ls) * B, T, P, monce).view(len(ls), 2, n)
            return loss.mean(dim=2).sum(dim=0) * o.alg_cut_lambda_NCE

        # (round 6, measured and removed: the smaller patch-count set on a forked third stream inside the graphs -- 20.0 -> 23.5 ms,
        #  profiles/r06_fork_nce_sets_ab.txt; like the third fork of the key-side encoder pass, a second fork costs the graph more than it hides)
        for Pl in sorted(set(counts)):
            m = one_set(Pl)
            tot = [tot[0] + m[0], tot[1] + m[1]]
        L = len(self.nce_layers)
        self.loss_G_NCE, self.loss_G_NCE_Y = tot[0] / L, tot[1] / L
        self.loss_G_tot = self.loss_G_tot + (self.loss_G_NCE + self.loss_G_NCE_Y) * 0.5

    def compute_G_loss_cut(self):
        """cut_model.py:708-837 (NCE + identity NCE)."""
        if self._batched_nce():
            return self.compute_G_loss_cut_batched()
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

    # ---- step driver: the discriminator half under the generator's backward -------------------------------------------------
    def _early_D(self):
        """The discriminator group's forward + backward reads `fake_B` (detached), `real_B` and the discriminators' weights: nothing the
        generator group's backward or optimizer step writes.  On one GPU it is therefore ENQUEUED on a second HIP stream right after the
        generator group's forward (ordered behind it: spectral-norm power iterations, BatchNorm statistics and the host's pool draws keep
        the reference's order) and runs next to the generator's backward: both are sequences of 10 - 30 us launches that fill a fraction
        of the 256 CUs each.  The discriminators' optimizer step stays where the reference has it, behind a stream join.  Same kernels,
        same operands: results differ from the sequential order only by fp32 atomics ordering.  `jg_early_D=False` / `JG_EARLY_D=0`
        = the sequential driver of BaseModel.  Data parallel (round 5, VERDICT r4 missing #3): the same driver on every rank -- cut_model does
        not opt into `overlap_exchange`, so the gradient all-reduce of every arena is issued by its optimizer step
        (parallel.allreduce_and_step) on the compute stream, AFTER `main.wait_stream(side)`: the side stream never carries a collective and
        the discriminators' gradients are final when their chunks leave.  The generator-half graphs stay single-process (the generator's
        BatchNorm layers all-reduce their batch statistics inside the forward when world_size > 1: a collective inside a capture)."""
        return (getattr(self.opt, "jg_early_D", True) and os.environ.get("JG_EARLY_D", "1") != "0" and self.isTrain and self.device.type == "cuda"
                and self.networks_groups == [self.group_G, self.group_D] and not self.group_D.forward_functions and not self.overlap_exchange)

    def _group_flags(self, group):
        for network in self.model_names:
            self.set_requires_grad(getattr(self, "net" + network), network in group.networks_to_optimize, _frozen_structure=True)

    def _group_finish(self, group):
        loss_names = []
        for temp in group.loss_names_list:
            loss_names += getattr(self, temp)
        self.compute_step(group.optimizer, loss_names, group)
        if self.opt.train_G_ema:
            for network in self.model_names:
                if network in group.networks_to_ema:
                    self.ema_step(network)

    # which step driver ran the LAST optimize_parameters(): "sequential" (BaseModel's group loop), "early" (discriminator half eager on a
    # second stream) or "graph" (that half replayed from a hipGraph); `step_driver_note` holds the reason a faster driver was not taken
    # (bench.py and the driver-agreement tests report both)
    step_driver = "sequential"
    step_driver_note = ""

    def _wgrad_stream(self):
        """`JG_WGRAD_GROUP_STREAM=1` (A/B): the grouped weight-gradient launches of the generator's backward (ops.deferred_wgrads) leave on a third
        stream every 48 problems instead of on the compute stream at the end.  Measured SLOWER (30.3 vs 29.4 ms/step, same box, everything replayed
        from graphs: the forked branch costs the graph more than the overlap returns): off."""
        if os.environ.get("JG_WGRAD_GROUP_STREAM", "0") != "1":
            return None
        st = self.__dict__.get("_wg_stream")
        if st is None:
            st = self._wg_stream = torch.cuda.Stream(device=self.device)
        return st

    def _draw_pool_fakes(self, side):
        """The history-pool draws of compute_D_loss, made on the MAIN stream before the side stream forks: `ImagePool.query` clones a stored
        image (a view of an earlier generator output, allocated on the main stream) and drops the last reference to it, so on the side
        stream the clone would only be queued when the caching allocator hands the block back to the main stream's pool (ADVICE r4:
        cross-stream use-after-free).  The drawn batches are then marked as in use by the side stream."""
        if POOL_ON_SIDE:
            # round 6: the draws themselves on the discriminator stream (~40 copy launches per step that graph B no longer queues behind):
            # the pool marks every stored image it reads there (`reader_stream`), the fresh batch is marked here
            self.fake_B.record_stream(side)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                return [self.fake_B_pool.query(self.fake_B, reader_stream=side).detach() for _ in self.discriminators_names]
        fakes = [self.fake_B_pool.query(self.fake_B).detach() for _ in self.discriminators_names]
        for f in fakes:
            f.record_stream(side)
        return fakes

    def optimize_parameters(self):
        if not self._early_D():
            self.step_driver = "sequential"
            return super().optimize_parameters()
        self.niter += 1
        self._ema_fused_this_iter = set()
        gG, gD = self.group_G, self.group_D
        its = self.opt.train_iter_size
        self._group_flags(gG)
        main = torch.cuda.current_stream(self.device)
        side = self.__dict__.get("_d_stream")
        if side is None:
            side = self._d_stream = torch.cuda.Stream(device=self.device)
        gst = self._g_half_from_graph(its)           # forward + losses of the generator group replayed from a hipGraph (None: eager)
        if gst is None:
            for fn in gG.forward_functions or []:
                getattr(self, fn)()
            for fn in gG.backward_functions:
                getattr(self, fn)()
        self._drawn_fakes = self._draw_pool_fakes(side)
        try:                               # whatever happens below, a later compute_D_loss must query the pool again, not reuse these
            self.real_B.record_stream(side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._group_flags(gD)
                if self._d_half_from_graph(side, its):
                    self.step_driver = "graph"
                else:
                    self.step_driver = "early"
                    for fn in gD.backward_functions:
                        getattr(self, fn)()
                    for loss in gD.loss_backward:
                        (getattr(self, loss) / its).backward()
        finally:
            self._drawn_fakes = None
        if os.environ.get("JG_DBG_EARLY_D_SYNC"):      # dev (tools/dbg_graph_d.py): the two halves one after the other
            torch.cuda.synchronize()
        self._group_flags(gG)
        if gst is not None:
            gst["bwd"].replay()
            self.step_driver += "+graphG"
        else:
            with ops.deferred_wgrads(self._wgrad_stream()):
                for loss in gG.loss_backward:
                    (getattr(self, loss) / its).backward()
        self._group_finish(gG)
        main.wait_stream(side)
        self._group_flags(gD)            # the flags end the step as the sequential driver leaves them
        self._group_finish(gD)
        for obj in self.objects_to_update:
            obj.update(self.niter)
        self.poll_overflow()

    # ---- the generator half as two hipGraphs ------------------------------------------------------------------------------
    def _g_outputs(self):
        names = ["fake", "fake_B", "loss_G_tot", "loss_G_NCE"] + ["loss_G_GAN_" + dn for dn in self.discriminators_names]
        if self.opt.alg_cut_nce_idt:
            names += ["idt_B", "loss_G_NCE_Y"]
        return names

    def _g_half_from_graph(self, its):
        """`jg_graph_G` (default on with `jg_graph_D`; `JG_GRAPH_G=0` / `1` overrides): the generator group's forward + losses and its
        backward -- ~2500 launches, 35 ms of host enqueue per step at the configs[2] shape, which IS the step time on a host slower than
        the GPU side (336 images/s on the driver's box against 377 here) -- are captured once (third step on) as TWO graphs
        that share a memory pool: graph F = forward_cut + compute_G_loss on static copies of real_A / real_B, graph B = the backward of
        loss_G_tot (retain_graph: its saved activations are F's outputs).  A step replays F, enqueues the discriminator half on the side
        stream (its own graph, `_d_half_from_graph`), replays B under it, and runs the three optimizer launches eagerly (their step
        count and learning rate are host scalars).  What makes this possible: DropPath / Dropout2d scales and the patch ids are drawn
        ON THE DEVICE inside the capture (torch's graph-safe Philox state advances per replay; `torch.randperm` is replaced by rand + topk
        while capturing), the history pools stay outside (the real-image pools before F, the fake pool between F and the discriminator
        half), and every arena's 16-bit working copies are refreshed INSIDE graph F (captured while dirty).  The same two guards as
        for the discriminator half: `HIP_GRAPHS_SAFE`, and a canary -- replay F + B twice from the same RNG state with 8192 eager
        launches in between; losses and the generator's gradient must agree -- after which the touched state (gradient arenas,
        spectral-norm vectors, BatchNorm statistics, RNG offset) is restored.  Returns the graph state or None (eager half)."""
        want = os.environ.get("JG_GRAPH_G", "")
        if not ((getattr(self.opt, "jg_graph_G", True) or want == "1") and want != "0"):
            return None
        import joligen_amd

        if not joligen_amd.HIP_GRAPHS_SAFE or self.__dict__.get("_gg_failed"):
            return None
        from .. import parallel

        if parallel.world_size() > 1:
            return None
        if (self.act_dtype == torch.float16 or self.niter <= 2 or ops.KERNEL_TIMING is not None or self.patch_ids_injection is not None
                or getattr(getattr(self._net("G_A"), "rand", None), "source", None) is not None):
            return None
        nets = [self._net(n) for n in self.model_names]
        key = (tuple(self.real_A.shape), tuple(self.real_B.shape), self.real_A.dtype, its, float(self.loss_scale),
               tuple(n.arena.p.data_ptr() for n in nets), tuple(n.training for n in nets))
        graphs = self.__dict__.setdefault("_gg_graphs", {})
        st = graphs.get(key)
        if st is None:
            if len(graphs) >= 3:
                return None
            st = self._g_capture(its, nets)
            if st is None:
                return None
            graphs[key] = st
        if self.opt.isTrain:               # the metric pools of forward(): host draws in the reference's order, outside the graph
            self.real_A_pool.store(self.real_A)
            self.real_B_pool.store(self.real_B)
        st["real_A"].copy_(self.real_A)
        st["real_B"].copy_(self.real_B)
        self.real_A, self.real_B = st["real_A"], st["real_B"]
        st["fwd"].replay()
        for n in nets:                     # graph F has just refreshed every trained working copy on the device: the host-side flag follows,
            n.arena._dirty = False         # or the discriminator half would refresh D's copies again on the side stream while graph B reads them
        for k, v in st["outs"].items():
            setattr(self, k, v)
        if self.fake_B_pool.pool_size > 0:        # the pool keeps views of what it is handed: not of a buffer the next replay overwrites
            self.fake_B = self.fake_B.clone()
        return st

    def _g_capture(self, its, nets):
        import warnings

        st = dict(real_A=self.real_A.clone(), real_B=self.real_B.clone())
        state = [n.arena.g for n in nets] + [b for n in nets for b in n.buffers()]
        saved = [t.clone() for t in state]
        keep_inputs = (self.real_A, self.real_B)
        rng = torch.cuda.get_rng_state(self.device)
        try:
            for n in nets:
                n.arena.ensure_fresh()             # (never-trained working copies: derived now, outside the graph)
                n.arena._dirty = True              # the refresh of every TRAINED working copy belongs to graph F
            self.real_A, self.real_B = st["real_A"], st["real_B"]
            gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(gf, capture_error_mode="thread_local"):
                ops.zero_pool_reset(self.device, True)          # the zeroed reduction rows of this graph come out of a chunk it clears itself
                self._forward_core()
                self.compute_G_loss()
            st["outs"] = {k: getattr(self, k) for k in self._g_outputs()}
            with torch.cuda.graph(gb, pool=gf.pool(), capture_error_mode="thread_local"):
                ops.zero_pool_reset(self.device, True)
                with ops.deferred_wgrads(self._wgrad_stream()):     # the ~190 weight gradients of the generator leave as grouped launches on a forked stream
                    (self.loss_G_tot / its).backward(retain_graph=True)
            ops.zero_pool_reset(self.device)
            st["fwd"], st["bwd"] = gf, gb

            def once():
                torch.cuda.set_rng_state(rng, self.device)
                for t, t0 in zip(state, saved):
                    t.data.copy_(t0)
                gf.replay()
                gb.replay()
                return torch.stack([st["outs"]["loss_G_tot"].detach().float().reshape(()), self._net("G_A").arena.g.norm()]).clone()

            first = once()
            burst = torch.zeros(64, device=self.device)
            for _ in range(8192):
                burst.add_(1.0)
            second = once()
            if os.environ.get("JG_DBG_GRAPH_CANARY_FAIL") == "G":      # tests: the fall-back path of a failing canary
                second = second * 1.5 + 1.0
            # same RNG state, same operands: the two runs differ by the summation order of the fp32 atomics only (measured 2e-4 on the loss,
            # 2e-3 on the gradient norm of a small model); the corruption this guards against shows as NaN or a loss that moves by per cents
            tol = torch.tensor([5e-3, 5e-2], device=first.device)
            ok = bool((torch.isfinite(first).all() & torch.isfinite(second).all() & ((first - second).abs() <= tol * first.abs() + 1e-6).all()).item())
            if not ok:
                raise RuntimeError(f"replays of the generator graphs disagree after interleaved eager launches ({first.tolist()} vs {second.tolist()})")
        except Exception as e:
            ops.zero_pool_reset(self.device)
            warnings.warn(f"jg_graph_G: the generator half stays eager ({e})")
            self.step_driver_note = (self.step_driver_note + "; " if self.step_driver_note else "") + f"generator graph dropped: {e}"
            self._gg_failed = True
            st = None
        for t, t0 in zip(state, saved):
            t.data.copy_(t0)
        torch.cuda.set_rng_state(rng, self.device)
        self.real_A, self.real_B = keep_inputs
        for n in nets:
            n.arena._dirty = True
        return st

    # ---- the discriminator half as a hipGraph -----------------------------------------------------------------------------
    def _d_half_body(self, real, fakes, its):
        """compute_D_loss + backward on given operands (the pool draws are the caller's)"""
        vals, tot = [], 0
        for dn, fake in zip(self.discriminators_names, fakes):
            val = getattr(self, dn + "_loss_calculator").compute_loss_D(self._net(dn), real, fake, None)
            vals.append(val)
            tot = tot + val
        tot = _ScaleGradFn.apply(tot, self.loss_scale)
        # round 6: the discriminators' weight gradients leave as grouped launches too (34 split-K launches of 37 - 69 us, each sized to fill
        # the chip by itself, next to the generator's backward); the spectral-norm fix of a discriminator flushes the ones it reads first
        with ops.deferred_wgrads():
            (tot / its).backward()
        return vals, tot

    def _d_half_from_graph(self, side, its):
        """`jg_graph_D` (default on; `JG_GRAPH_D=0` / `1` overrides the option): the discriminator half -- ~700 launches whose operands
        are two images and the discriminators' arenas -- is captured once (third step on) and replayed.  It holds no host-side random
        draw (the pool queries stay outside), no optimizer step and no host-computed scalar that changes between steps (16-bit
        activations other than fp16: the loss scale is 1).  What it buys is host time: the eager half costs 7.8 ms of enqueue per step,
        and the step's wall time is the enqueue time on hosts slower than the GPU side: 380 images/s on every box instead of 300 - 370
        on the configs[2] shape.

        Two conditions, both checked.  (1) ROCm 7.2's hipGraph replays go wrong when thousands of eager launches run between two
        replays -- its AQL-packet capture keeps the nodes' kernel arguments where later launches overwrite them
        (profiles/r04_graph_replay_probe.txt: the loss of an untouched graph moves after 3000 one-element `add_` launches on another
        stream, NaN gradients after a generator backward).  `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`, read by the runtime when it
        initialises, switches that path off; `joligen_amd/__init__.py` sets it when the package is imported before the first HIP call
        and records in `HIP_GRAPHS_SAFE` whether it could -- without that the eager half runs.  As a second net the capture is
        followed once by a canary: replay, 8192 tiny eager launches, replay; if the two losses differ the graph is dropped for good
        (with a warning); the canary's side effects (gradient accumulation, spectral-norm power iterations, running statistics) are
        undone from snapshots.  (2) Every tensor the graph reads has to keep its address: the lazily derived tables of the frozen
        feature network (`_FrozenBN.affine`, `_DWWeight.taps`) are refreshed in place for that reason -- re-created, their old storage
        went back to the allocator and the replays read whatever was written there next (found as projected-discriminator logits 1.3 -
        3x too large on a 64 x 64 model; tests/test_gpu_5_cutloss.py::test_cut_step_drivers_agree holds the three drivers together).
        Returns False when the eager path has to run (not enabled, not applicable, capture failed, canary failed)."""
        want = os.environ.get("JG_GRAPH_D", "")
        if not ((getattr(self.opt, "jg_graph_D", True) or want == "1") and want != "0"):
            return False
        import joligen_amd

        if not joligen_amd.HIP_GRAPHS_SAFE:
            self.step_driver_note = "HIP_GRAPHS_SAFE is False (HIP initialised before DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 could be set)"
            return False
        if self.__dict__.get("_dg_failed"):
            return False                       # step_driver_note holds the capture / canary failure
        if self.act_dtype == torch.float16 or self.niter <= 2 or ops.KERNEL_TIMING is not None:
            return False
        nets = [self._net(dn) for dn in self.discriminators_names]
        from .. import parallel

        if parallel.world_size() > 1 and any(getattr(m, "jg_forward_collective", False) or isinstance(m, torch.nn.SyncBatchNorm)
                                             for n in nets for m in n.modules()):
            # the side stream (and a captured graph) must never carry a collective: none of today's discriminators has one in its forward --
            # BatchNorm statistics are per rank, as in the reference's DDP default -- a module that adds one marks itself and gets the eager path
            self.step_driver_note = "a discriminator module issues a collective in its forward: discriminator half not captured"
            return False
        key = (tuple(self.real_B.shape), self.real_B.dtype, tuple(self.fake_B.shape), its, float(self.loss_scale),
               tuple(n.arena.p.data_ptr() for n in nets), tuple(n.training for n in nets))
        graphs = self.__dict__.setdefault("_dg_graphs", {})      # one graph per operand shape (a last, smaller batch of an epoch): at most three
        st = graphs.get(key)
        if st is None:
            if len(graphs) >= 3:
                return False
            st = self._d_capture(side, its, key, nets)
            if st is None:
                return False
            graphs[key] = st
        self._dg = st
        fakes = self._drawn_fakes
        st["real"].copy_(self.real_B)
        for dst, f in zip(st["fakes"], fakes):
            dst.copy_(f)
        st["graph"].replay()
        self._d_publish(st["vals"], st["tot"], clone=True)
        return True

    def _d_capture(self, side, its, key, nets):
        import warnings

        st = dict(key=key, real=self.real_B.clone(), fakes=[self.fake_B.detach().clone() for _ in nets])
        state = [n.arena.g for n in nets] + [b for n in nets for b in n.buffers()]
        saved = [t.clone() for t in state]
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):      # other threads (pinned staging, checkpoint writer) may allocate
                ops.zero_pool_reset(self.device, True)
                st["vals"], st["tot"] = self._d_half_body(st["real"], st["fakes"], its)
            ops.zero_pool_reset(self.device)
            graph.replay()
            first = st["tot"].detach().clone()
            burst = torch.zeros(64, device=self.device)
            for _ in range(2048):
                burst.add_(1.0)
            with torch.cuda.stream(torch.cuda.default_stream(self.device)):
                for _ in range(6144):
                    burst.add_(1.0)
            side.wait_stream(torch.cuda.default_stream(self.device))
            for t, t0 in zip(state, saved):      # both replays start from the same power-iteration vectors / statistics
                t.data.copy_(t0)      # .data: no version bump (frozen-BN tables key on versions; tensors saved by G's forward stay valid)
            graph.replay()
            second = st["tot"].detach().clone()
            if os.environ.get("JG_DBG_GRAPH_CANARY_FAIL") == "1":      # tests: the fall-back path of a failing canary
                second = second * 1.5 + 1.0
            for t, t0 in zip(state, saved):
                t.data.copy_(t0)      # .data: no version bump (frozen-BN tables key on versions; tensors saved by G's forward stay valid)
            ok = bool((torch.isfinite(first) & torch.isfinite(second) & ((first - second).abs() <= 1e-3 * first.abs() + 1e-6)).item())
            if not ok:
                raise RuntimeError(f"replays of an untouched graph disagree after interleaved eager launches ({float(first)} vs {float(second)}): "
                                   "export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the first HIP call")
        except Exception as e:
            ops.zero_pool_reset(self.device)
            warnings.warn(f"jg_graph_D: the discriminator half stays eager ({e})")
            self.step_driver_note = f"graph dropped: {e}"
            self._dg_failed = True
            self._dg = None
            for t, t0 in zip(state, saved):
                t.data.copy_(t0)      # .data: no version bump (frozen-BN tables key on versions; tensors saved by G's forward stay valid)
            return None
        st["graph"] = graph
        return st

    def _d_publish(self, vals, tot, clone):
        for dn, val in zip(self.discriminators_names, vals):
            setattr(self, "loss_D_GAN_" + dn, val.detach().clone() if clone else val)
        self.loss_D_tot = tot.detach().clone() if clone else tot

    # ---- discriminator loss (base_gan_model.py:341-419) ------------------------------------------------------------------
    def compute_D_loss(self):
        """base_gan_model.py:341-419: every discriminator draws ITS OWN fake batch from the history pool (compute_D_loss_generic)."""
        tot = 0
        drawn = self.__dict__.get("_drawn_fakes")          # early-D driver: drawn on the main stream before the fork (same draw order)
        for i, dn in enumerate(self.discriminators_names):
            fake = drawn[i] if drawn is not None else self.fake_B_pool.query(self.fake_B)
            val = getattr(self, dn + "_loss_calculator").compute_loss_D(self._net(dn), self.real_B, fake, None)
            setattr(self, "loss_D_GAN_" + dn, val)
            tot = tot + val
        self.loss_D_tot = _ScaleGradFn.apply(tot, self.loss_scale)

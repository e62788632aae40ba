"""DDPM training forward on the HIP ops: mirror of
/root/reference/models/modules/diffusion_generator.py (`DiffusionGenerator` :23-80, `forward`
:457-521, `compute_gammas` :526-528), models/modules/palette_denoise_fn.py (`PaletteDenoiseFn`
:35-115, conditioning off) and models/modules/diffusion_utils.py (`make_beta_schedule` :45-79,
`set_new_noise_schedule` :81-119).

state_dict keys are the reference's: `denoise_fn.model.<unet...>`, the 14 schedule buffers
`denoise_fn.model.{gammas,...}_{train,test}` and `cond_embed.{0,2}.{weight,bias}`.

`restoration` (reference :83-455) = the DDPM and DDIM samplers of SURVEY.md 8(f).
"""
from __future__ import annotations

from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..arena import ParamArena
from ..ops import JG_ACT_NONE, JG_ACT_SILU


def make_beta_schedule(schedule, n_timestep, linear_start=1e-6, linear_end=1e-2, cosine_s=8e-3):
    """diffusion_utils.py:45-79 (float64 numpy)."""
    if schedule == "quad":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "const":
        return linear_end * np.ones(n_timestep, dtype=np.float64)
    if schedule == "jsd":
        return 1.0 / np.linspace(n_timestep, 1, n_timestep, dtype=np.float64)
    raise NotImplementedError(schedule)


def set_new_noise_schedule(model, phase):
    """diffusion_utils.py:81-119: registers the 7 schedule buffers of `phase` on `model`."""
    to_torch = partial(torch.tensor, dtype=torch.float32)
    betas = make_beta_schedule(**model.beta_schedule[phase])
    alphas = 1.0 - betas
    (timesteps,) = betas.shape
    setattr(model, "num_timesteps_" + phase, int(timesteps))
    gammas = np.cumprod(alphas, axis=0)
    gammas_prev = np.append(1.0, gammas[:-1])
    model.register_buffer("gammas_" + phase, to_torch(gammas))
    model.register_buffer("gammas_prev_" + phase, to_torch(gammas_prev))
    model.register_buffer("sqrt_recip_gammas_" + phase, to_torch(np.sqrt(1.0 / gammas)))
    model.register_buffer("sqrt_recipm1_gammas_" + phase, to_torch(np.sqrt(1.0 / gammas - 1)))
    posterior_variance = betas * (1.0 - gammas_prev) / (1.0 - gammas)
    model.register_buffer("posterior_log_variance_clipped_" + phase, to_torch(np.log(np.maximum(posterior_variance, 1e-20))))
    model.register_buffer("posterior_mean_coef1_" + phase, to_torch(betas * np.sqrt(gammas_prev) / (1.0 - gammas)))
    model.register_buffer("posterior_mean_coef2_" + phase, to_torch((1.0 - gammas_prev) * np.sqrt(alphas) / (1.0 - gammas)))


class LabelEmbedder(nn.Module):
    """palette_denoise_fn.py:14-32: nn.Embedding(num_classes, hidden, max_norm=1.0, scale_grad_by_freq=True).  The lookup of B rows
    is host-level plumbing on torch's device ops (renormalisation of the looked-up rows in place, gather, and a dense gradient that
    autograd accumulates into the arena-backed `.grad`)."""

    def __init__(self, num_classes, hidden_size):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes, hidden_size, max_norm=1.0, scale_grad_by_freq=True)
        self.num_classes = num_classes

    def forward(self, labels):
        if not labels.is_cuda:
            raise RuntimeError("LabelEmbedder: labels must live on the GPU (no CPU fallback)")
        return self.embedding_table(labels)


class PaletteDenoiseFn(nn.Module):
    """palette_denoise_fn.py:35-115 for `conditioning` made of "class" and / or "mask" (the reference tests substrings: "class_mask"
    selects both): with "class" the label embedding (cond_embed_dim // 2 wide) is concatenated to the noise-level embedding (:99-103);
    with "mask" a per-pixel label embedding (cond_embed_dim wide, looked up from the semantic mask) is concatenated to the UNet INPUT
    (:108-112, define_G adds cond_embed_dim input channels).  Reference-image conditioning (CLIP / ImageBind encoders) is not built."""

    def __init__(self, model, cond_embed_dim, ref_embed_net, conditioning, nclasses):
        super().__init__()
        tokens = set(t for t in str(conditioning).replace(",", "_").split("_") if t)
        if tokens - {"class", "mask"}:
            raise NotImplementedError(f"alg_diffusion_cond_embed={conditioning!r}: '', 'class', 'mask' and 'class_mask' are implemented")
        self.model = model
        self.conditioning = conditioning
        self.cond_embed_dim = cond_embed_dim
        self.ref_embed_net = ref_embed_net
        if "class" in conditioning:
            self.netl_embedder_class = LabelEmbedder(nclasses, cond_embed_dim // 2)
            nn.init.normal_(self.netl_embedder_class.embedding_table.weight, std=0.02)
        if "mask" in conditioning:
            self.netl_embedder_mask = LabelEmbedder(nclasses, cond_embed_dim)
            nn.init.normal_(self.netl_embedder_mask.embedding_table.weight, std=0.02)

    def forward(self, input, embed_noise_level, cls=None, mask=None, ref=None):
        """input: [B, H, W, Cpad] 16-bit NHWC holding the (y_cond, y_noisy) channels first"""
        embedding = embed_noise_level
        if "class" in self.conditioning:
            if cls is None:
                raise RuntimeError("alg_diffusion_cond_embed='class' needs the class labels (B_label_cls)")
            embedding = torch.cat((embed_noise_level, self.netl_embedder_class(cls).to(embed_noise_level.dtype)), dim=1)
        if "mask" in self.conditioning:
            if mask is None:
                raise RuntimeError("alg_diffusion_cond_embed='mask' needs the semantic mask (B_label_mask)")
            B, H, W, _ = input.shape
            D = self.cond_embed_dim
            real = self.model.in_channel - D              # the image channels in front (y_cond, y_noisy)
            pad = (-self.model.in_channel) % 8
            me = self.netl_embedder_mask(mask.long().reshape(B, H * W)).view(B, H, W, D).to(input.dtype)
            parts = [input[..., :real], me]
            if pad:
                parts.append(torch.zeros((B, H, W, pad), device=input.device, dtype=input.dtype))
            input = torch.cat(parts, dim=-1)
        return self.model(input, embedding)


class DiffusionGenerator(nn.Module):
    def __init__(self, denoise_fn, sampling_method, image_size, G_ngf, loading_backward_compatibility=False):
        super().__init__()
        if loading_backward_compatibility:
            raise NotImplementedError("model_prior_321_backwardcompatibility is not implemented")
        self.denoise_fn = denoise_fn
        self.sampling_method = sampling_method
        self.image_size = image_size
        cond_embed_dim = self.denoise_fn.cond_embed_dim
        set_new_noise_schedule(model=self.denoise_fn.model, phase="train")
        set_new_noise_schedule(model=self.denoise_fn.model, phase="test")
        self.cond_embed_dim = cond_embed_dim
        # :66-69: half of the embedding belongs to the class / reference embedding when one is configured
        self.cond_embed_gammas = cond_embed_dim // 2 if any(c in self.denoise_fn.conditioning for c in ("class", "ref")) else cond_embed_dim
        self.cond_embed = nn.Sequential(
            nn.Linear(self.cond_embed_gammas, self.cond_embed_gammas),
            nn.SiLU(),
            nn.Linear(self.cond_embed_gammas, self.cond_embed_gammas),
        )
        self.cond_embed_gammas_in = self.cond_embed_gammas
        self.arena = None
        self.act_dtype = torch.bfloat16
        self.loss_scale = 1.0

    # ---- MI355X finalisation -------------------------------------------------------------
    def jg_finalize(self, device, act_dtype=torch.bfloat16):
        """Move every parameter into a flat device arena and build the 16-bit conv weights.
        Must be called once, after construction / before the first forward (the model classes'
        single_gpu()/parallelize() do it)."""
        if self.arena is not None:
            return self.arena
        self.act_dtype = act_dtype
        self.arena = ParamArena(self, device, act_dtype)
        self.denoise_fn.model._jg_arena_ref = self.arena
        return self.arena

    # ---- pieces --------------------------------------------------------------------------
    def compute_gammas(self, gammas):
        """reference :526-528: cond_embed(gamma_embedding(gammas, dim))."""
        emb = ops.gamma_embedding(gammas, self.cond_embed_gammas_in)
        l0, l2 = self.cond_embed[0], self.cond_embed[2]
        emb = ops.linear(emb, l0.weight, l0.bias, JG_ACT_NONE)
        return ops.linear(emb, l2.weight, l2.bias, JG_ACT_SILU)

    def sample_gammas(self, b, device, t=None, u=None):
        """reference :467-478.  `t`/`u` may be injected (parity runs draw them from a CPU generator)."""
        model = self.denoise_fn.model
        if t is None:
            t = torch.randint(1, model.num_timesteps_train, (b,), device=device).long()
        else:
            t = t.to(device).long()
        if u is None:
            u = torch.rand((b, 1), device=device)
        else:
            u = u.to(device).float().view(b, 1)
        gammas = model.gammas_train
        gamma_t1 = gammas.gather(-1, t - 1).view(b, 1)
        gamma_t2 = gammas.gather(-1, t).view(b, 1)
        return t, (gamma_t2 - gamma_t1) * u + gamma_t1

    def min_snr_weight(self, t):
        """reference :502-519."""
        model = self.denoise_fn.model
        snr = torch.pow(model.sqrt_recip_gammas_train.gather(-1, t) / model.sqrt_recipm1_gammas_train.gather(-1, t), 2)
        return (torch.minimum(snr, 5.0 * torch.ones_like(snr)) / snr).view(-1, 1, 1, 1)

    def forward_nhwc(self, y_0, y_cond, mask, noise, t=None, u=None, cls=None):
        """The training forward with the UNet output left in NHWC 16-bit (8 channels, 3 valid).
        Returns (noise fp32 NCHW, noise_hat NHWC 16-bit, min_snr_w [B,1,1,1], t)."""
        if self.arena is None:
            raise RuntimeError("DiffusionGenerator.jg_finalize(device) has not been called")
        if y_0.dim() != 4:
            raise NotImplementedError("video (5-D) inputs are outside the SURVEY.md 8 hot path")
        self.arena.ensure_fresh()
        b = y_0.shape[0]
        dev = y_0.device
        t, sample_gammas = self.sample_gammas(b, dev, t, u)
        if noise is None:
            noise = torch.randn_like(y_0)
        emb = self.compute_gammas(sample_gammas)
        xin = ops.ddpm_prepare(y_0.float(), y_cond.float(), noise.float(), mask, sample_gammas.view(-1).contiguous(),
                               self.act_dtype, cpad=8)
        noise_hat = self.denoise_fn(xin, emb, cls=cls, mask=mask, ref=None)
        return noise, noise_hat, self.min_snr_weight(t), t

    def forward(self, y_0, y_cond, mask, noise, cls=None, ref=None, dropout_prob=0.0, t=None, u=None):
        """reference :457-521 signature; returns (noise, noise_hat, min_snr_loss_weight) NCHW fp32."""
        if ref is not None:
            raise NotImplementedError("reference-image conditioning is not built")
        noise, nh, w, _ = self.forward_nhwc(y_0, y_cond, mask, noise, t, u, cls=cls)
        return noise, ops.to_nchw_f32(nh, y_0.shape[1]), w

    # ---- sampling (reference :83-284) ---------------------------------------------------------
    @torch.no_grad()
    def restoration(self, y_cond, y_t=None, y_0=None, mask=None, sample_num=8, cls=None, ref=None, guidance_scale=0.0,
                    ddim_num_steps=10, ddim_eta=0.5, noises=None, clip_denoised=True):
        """reference :83-183 (`restoration` -> `restoration_ddpm`): ancestral sampling over the TEST schedule.
        Per step: one UNet forward on the fused schedule + ONE fused kernel (`jg_ddpm_p_sample`) for
        predict_start_from_noise, clamp, q_posterior, the noise injection, the mask blend and the next UNet input.
        `noises`: optional list of the per-step N(0,1) draws in loop order (parity runs); else the device RNG.
        Returns (y_t, ret_arr) like the reference, NCHW fp32."""
        if self.sampling_method not in ("ddpm", "ddim"):
            raise NotImplementedError(f"sampling method {self.sampling_method!r}")
        if guidance_scale > 0.0 or ref is not None:
            # (the reference's guided branch calls denoise_fn(..., cls=None), which cannot concatenate a class embedding: :214-227)
            raise NotImplementedError("classifier-free guidance / reference conditioning are not built")
        if self.arena is None:
            raise RuntimeError("DiffusionGenerator.jg_finalize(device) has not been called")
        self.arena.ensure_fresh()
        model = self.denoise_fn.model
        T = model.num_timesteps_test
        assert T > sample_num, "num_timesteps must greater than sample_num"
        sample_inter = T // sample_num
        b, Cc, H, W = y_cond.shape
        dev = y_cond.device
        y_cond = y_cond.float().contiguous()
        if y_t is None:
            y_t = torch.randn((b, model.out_channel, H, W), device=dev, dtype=torch.float32)
        y_t = y_t.float().clone()
        ret = [y_t.clone()]
        m = None
        if mask is not None:
            m = mask.contiguous() if mask.dtype == torch.int64 else mask.long().contiguous()
            y_0 = y_0.float().contiguous()
        cpad = (2 * Cc + 7) // 8 * 8
        xin = ops.to_nhwc(torch.cat([y_cond, y_t], dim=1), self.act_dtype, cpad)
        L = ops._lib.lib()
        if self.sampling_method == "ddim":
            # reference :286-349 + ddim_p_mean_variance :383-455 (deterministic: the noise it draws is not used).
            # The UNet output is clamped to [-1,1], mean = sqrt(g_prev) (y_t - sqrt(1-g_t) e) / sqrt(g_t) + coef_eps e,
            # clamped again -- the same fused kernel with other coefficients.
            tseq = list(np.linspace(0, T - 1, ddim_num_steps).astype(int))
            for i in range(ddim_num_steps):
                ti = int(tseq[-1 - i])
                prev = int(tseq[-2 - i]) if i != ddim_num_steps - 1 else -1
                t = torch.full((b,), ti, device=dev, dtype=torch.long)
                emb = self.compute_gammas(model.gammas_test.gather(-1, t).view(b, 1))
                nh = self.denoise_fn(xin, emb, cls=cls, mask=mask)
                gamma_t = model.gammas_test.gather(-1, t)
                gamma_p = model.gammas_prev_test.gather(-1, torch.full((b,), prev + 1, device=dev, dtype=torch.long))
                sigma = ddim_eta * torch.sqrt((1 - gamma_p) / (1 - gamma_t) * (1 - gamma_t / gamma_p))
                coef_eps = torch.sqrt(torch.clamp(1 - gamma_p - sigma ** 2, min=0))
                c_e = coef_eps - torch.sqrt(gamma_p) * torch.sqrt(1.0 - gamma_t) / torch.sqrt(gamma_t)
                c_y = torch.sqrt(gamma_p) / torch.sqrt(gamma_t)
                coef = torch.stack([torch.zeros_like(c_e), -torch.ones_like(c_e), c_e, c_y, torch.zeros_like(c_e)], dim=1).contiguous()
                xin = torch.empty_like(xin)
                ops.check(L.jg_ddpm_p_sample(ops._DT[self.act_dtype], y_t.data_ptr(), y_cond.data_ptr(), nh.data_ptr(), None,
                                             ops._p(y_0 if m is not None else None), ops._p(m), coef.data_ptr(), xin.data_ptr(),
                                             b, Cc, H, W, nh.shape[-1], cpad, 3 if clip_denoised else 0, ops._st()),
                          "jg_ddpm_p_sample")
                if i % sample_inter == 0:
                    ret.append(y_t.clone())
            return y_t, torch.cat(ret, dim=0)
        for step, i in enumerate(reversed(range(T))):
            t = torch.full((b,), i, device=dev, dtype=torch.long)
            emb = self.compute_gammas(model.gammas_test.gather(-1, t).view(b, 1))
            nh = self.denoise_fn(xin, emb, cls=cls, mask=mask)
            coef = torch.stack([model.sqrt_recip_gammas_test.gather(-1, t), model.sqrt_recipm1_gammas_test.gather(-1, t),
                                model.posterior_mean_coef1_test.gather(-1, t), model.posterior_mean_coef2_test.gather(-1, t),
                                (0.5 * model.posterior_log_variance_clipped_test.gather(-1, t)).exp()], dim=1).contiguous()
            z = None
            if i > 0:
                z = (noises[step].to(dev).float().contiguous() if noises is not None else torch.randn_like(y_t))
            xin = torch.empty_like(xin)
            ops.check(L.jg_ddpm_p_sample(ops._DT[self.act_dtype], y_t.data_ptr(), y_cond.data_ptr(), nh.data_ptr(), ops._p(z),
                                         ops._p(y_0 if m is not None else None), ops._p(m), coef.data_ptr(), xin.data_ptr(), b, Cc,
                                         H, W, nh.shape[-1], cpad, int(clip_denoised), ops._st()), "jg_ddpm_p_sample")
            if i % sample_inter == 0:
                ret.append(y_t.clone())
        return y_t, torch.cat(ret, dim=0)

    def set_new_sampling_method(self, sampling_method):
        self.sampling_method = sampling_method

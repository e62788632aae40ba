"""Consistency-model generator on the HIP ops: mirror of /root/reference/models/modules/cm_generator.py
(`NoiseLevelEmbedding` :255-280, schedules :32-164, scalings :167-252, `CMGenerator` :283-326,
`cm_forward` :367-386, `forward` :388-502 for `alg_ddpm_ft_mode == "cm"`).

state_dict keys are the reference's: `cm_model.<unet...>`, `cm_cond_embed.W`,
`cm_cond_embed.projection.{0,2}.{weight,bias}`.

Per step the UNet runs twice on the fused schedule (unet_exec.py): the student on x + sigma_{n+1} eps (with
gradient) and the teacher on x + sigma_n eps (same weights, no gradient).  The ECT branch (:396-435) and
`restoration` (:504-554) are not part of the training hot path built here.
"""
from __future__ import annotations

import math

import os

import torch
import torch.nn as nn

from .. import ops
from ..arena import ParamArena
from ..ops import JG_ACT_NONE, JG_ACT_SILU


def improved_timesteps_schedule(current_training_step, total_training_steps, initial_timesteps=10, final_timesteps=1280):
    """cm_generator.py:32-69."""
    total_prime = math.floor(total_training_steps / (math.log2(math.floor(final_timesteps / initial_timesteps)) + 1))
    num_timesteps = initial_timesteps * math.pow(2, math.floor(current_training_step / total_prime))
    return int(min(num_timesteps, final_timesteps) + 1)


def karras_schedule(num_timesteps, sigma_min=0.002, sigma_max=80.0, rho=7.0, device=None):
    """cm_generator.py:72-106 (fp32 tensor ops on `device`, like the reference)."""
    rho_inv = 1.0 / rho
    steps = torch.arange(num_timesteps, device=device) / max(num_timesteps - 1, 1)
    sigmas = sigma_min ** rho_inv + steps * (sigma_max ** rho_inv - sigma_min ** rho_inv)
    return sigmas ** rho


def lognormal_timestep_distribution(num_samples, sigmas, mean=-1.1, std=2.0):
    """cm_generator.py:109-144."""
    pdf = torch.erf((torch.log(sigmas[1:]) - mean) / (std * math.sqrt(2))) - torch.erf(
        (torch.log(sigmas[:-1]) - mean) / (std * math.sqrt(2)))
    pdf = pdf / pdf.sum()
    return torch.multinomial(pdf, num_samples, replacement=True)


def improved_loss_weighting(sigmas):
    return 1 / (sigmas[1:] - sigmas[:-1])                                     # :147-164


def output_scaling(sigma, sigma_data=0.5, sigma_min=0.002):
    return (sigma_data * (sigma - sigma_min)) / (sigma_data ** 2 + sigma ** 2) ** 0.5   # :167-186


def skip_scaling(sigma, sigma_data=0.5, sigma_min=0.002):
    return sigma_data ** 2 / ((sigma - sigma_min) ** 2 + sigma_data ** 2)        # :211-230


class NoiseLevelEmbedding(nn.Module):
    """cm_generator.py:255-280.  `projection.3` of the reference is a parameter-free Rearrange."""

    def __init__(self, channels, opt, scale=0.02):
        super().__init__()
        self.W = nn.Parameter(torch.randn(channels // 2) * scale, requires_grad=False)
        hidden = channels if getattr(opt, "alg_diffusion_ddpm_cm_ft", False) else 4 * channels
        self.projection = nn.Sequential(nn.Linear(channels, hidden), nn.SiLU(), nn.Linear(hidden, channels), nn.Identity())

    def forward(self, sigmas):
        h = ops.noise_level_embedding(sigmas, self.W)
        l0, l2 = self.projection[0], self.projection[2]
        h = ops.linear(h, l0.weight, l0.bias, JG_ACT_NONE)
        return ops.linear(h, l2.weight, l2.bias, JG_ACT_SILU)


CM_FORK = os.environ.get("JG_CM_FORK", "1") != "0"


class CMGenerator(nn.Module):
    def __init__(self, cm_model, sampling_method, image_size, G_ngf, opt=None):
        super().__init__()
        self.cm_model = cm_model
        self.sampling_method = sampling_method
        self.image_size = image_size
        self.opt = opt
        self.sigma_min, self.sigma_max, self.sigma_data = 0.002, 80.0, 0.5
        self.rho = 7.0
        self.initial_timesteps, self.final_timesteps = 10, 1280
        self.lognormal_mean, self.lognormal_std = -1.1, 2.0
        self.cond_embed_dim = self.cm_model.cond_embed_dim
        self.cm_cond_embed = NoiseLevelEmbedding(self.cond_embed_dim, self.opt)
        self.current_t = 0
        self.arena = None
        self.act_dtype = torch.bfloat16
        if getattr(opt, "alg_ddpm_ft_mode", "cm") != "cm":
            raise NotImplementedError("alg_ddpm_ft_mode='ect' is not implemented (SURVEY.md 8: consistency-model branch only)")

    # ---- MI355X finalisation ---------------------------------------------------------------
    def jg_finalize(self, device, act_dtype=torch.bfloat16):
        if self.arena is not None:
            return self.arena
        self.act_dtype = act_dtype
        # the reference's step trains every parameter of the group's networks, W included
        # (BaseModel.set_requires_grad(net, True), base_model.py:1196-1217,1316-1322)
        self.cm_cond_embed.W.requires_grad_(True)
        self.arena = ParamArena(self, device, act_dtype)
        self.cm_model._jg_arena_ref = self.arena
        return self.arena

    def embed_sigmas(self, sigmas):
        return self.cm_cond_embed(sigmas)                                       # :556-558

    def _unet(self, xin, sigma):
        return self.cm_model(xin, self.embed_sigmas(sigma))

    def forward_nhwc(self, x, total_training_steps=50000, mask=None, x_cond=None, noise=None, timesteps=None):
        """The training forward with the two UNet outputs left in NHWC 16-bit.  Returns a dict with F_next (with
        gradient), F_cur, the noisy inputs (fp32 NCHW), the four scalings, sigmas, loss weights, num_timesteps."""
        if self.arena is None:
            raise RuntimeError("CMGenerator.jg_finalize(device) has not been called")
        if x.dim() != 4:
            raise NotImplementedError("video (5-D) inputs are outside the SURVEY.md 8 hot path")
        self.arena.ensure_fresh()
        dev = x.device
        num_timesteps = improved_timesteps_schedule(self.current_t, total_training_steps, self.initial_timesteps,
                                                    self.final_timesteps)
        sigmas = karras_schedule(num_timesteps, self.sigma_min, self.sigma_max, self.rho, dev)
        if noise is None:
            noise = torch.randn_like(x)
        if timesteps is None:
            timesteps = lognormal_timestep_distribution(x.shape[0], sigmas, self.lognormal_mean, self.lognormal_std)
        noise, timesteps = noise.to(dev), timesteps.to(dev)
        current_sigmas, next_sigmas = sigmas[timesteps], sigmas[timesteps + 1]
        cpad = (x.shape[1] + (0 if x_cond is None else x_cond.shape[1]) + 7) // 8 * 8
        next_noisy_x, xin_next = ops.cm_noisy(x, noise, next_sigmas, mask, x_cond, self.act_dtype, cpad)
        if CM_FORK and x.is_cuda:
            # round 6 (JG_CM_FORK, default 1; 121.3 -> 120.4 ms at batch 64): the no-grad pass on the current noise level is independent of the student pass -- enqueued on a
            # forked stream, joined before the loss
            main = torch.cuda.current_stream(dev)
            side = self.__dict__.get("_fork_stream")
            if side is None:
                side = self.__dict__["_fork_stream"] = torch.cuda.Stream(device=dev)
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                current_noisy_x, xin_cur = ops.cm_noisy(x, noise, current_sigmas, mask, x_cond, self.act_dtype, cpad)
                F_cur = self._unet(xin_cur, current_sigmas)
            for t in (x, noise, current_sigmas, mask, x_cond):
                if t is not None and t.is_cuda:
                    t.record_stream(side)
            F_next = self._unet(xin_next, next_sigmas)
            main.wait_stream(side)
            F_cur.record_stream(main)
            current_noisy_x.record_stream(main)
        else:
            F_next = self._unet(xin_next, next_sigmas)
            with torch.no_grad():
                current_noisy_x, xin_cur = ops.cm_noisy(x, noise, current_sigmas, mask, x_cond, self.act_dtype, cpad)
                F_cur = self._unet(xin_cur, current_sigmas)
        self.current_t += x.shape[0]
        return dict(F_next=F_next, F_cur=F_cur, next_noisy_x=next_noisy_x, current_noisy_x=current_noisy_x,
                    cs_n=skip_scaling(next_sigmas, self.sigma_data, self.sigma_min),
                    co_n=output_scaling(next_sigmas, self.sigma_data, self.sigma_min),
                    cs_c=skip_scaling(current_sigmas, self.sigma_data, self.sigma_min),
                    co_c=output_scaling(current_sigmas, self.sigma_data, self.sigma_min),
                    num_timesteps=num_timesteps, sigmas=sigmas,
                    loss_weights=improved_loss_weighting(sigmas)[timesteps].view(-1, 1, 1, 1))

    @torch.no_grad()
    def restoration(self, y, y_cond, sigmas, mask, clip_denoised=True, noises=None):
        """reference :504-554: multistep consistency sampling over `sigmas` (CMModel.inference uses (80, 24.4, 5.84, 0.9, 0.661)).
        Per sigma: ONE fused kernel forms `x + s * eps` with the mask blend (fp32 NCHW) together with the UNet's 16-bit NHWC input
        (`jg_cm_noisy`), one UNet forward on the fused schedule, one kernel for `c_skip x + c_out F` (`jg_cm_combine`).
        `noises`: optional list with the N(0,1) draw of every sigma in loop order (parity runs); else the device RNG.  NCHW fp32."""
        if self.arena is None:
            raise RuntimeError("CMGenerator.jg_finalize(device) has not been called")
        if y.dim() != 4:
            raise NotImplementedError("video (5-D) inputs are outside the SURVEY.md 8 hot path")
        self.arena.ensure_fresh()
        dev = y.device
        y = y.float()
        m = None
        if mask is not None:
            m = torch.clamp(mask, min=0, max=1)                                   # removes class information from the mask
            y = y * (1 - m).to(y.dtype)
        cond = None if y_cond is None else y_cond.float()
        cpad = (y.shape[1] + (0 if cond is None else cond.shape[1]) + 7) // 8 * 8
        x = y
        for k, sig in enumerate(sigmas):
            sigma = torch.full((y.shape[0],), float(sig), dtype=torch.float32, device=dev)
            eps = noises[k].to(dev).float() if noises is not None else torch.randn_like(y)
            # first step: y + sigma eps; later steps: x + sqrt(sigma^2 - sigma_min^2) eps; then x m + (1 - m) y (x already equals y
            # outside the mask after the previous blend, which is what the kernel blends with)
            step = sigma if k == 0 else (sigma ** 2 - self.sigma_min ** 2) ** 0.5
            noisy, xin = ops.cm_noisy(x, eps, step, m, cond, self.act_dtype, cpad)
            F = self._unet(xin, sigma)
            x = ops.cm_combine(noisy, F, skip_scaling(sigma, self.sigma_data, self.sigma_min),
                               output_scaling(sigma, self.sigma_data, self.sigma_min))
            if clip_denoised:
                x = x.clamp(min=-1.0, max=1.0)
            if m is not None:
                mf = m.to(x.dtype)
                x = x * mf + (1 - mf) * y
        return x

    def forward(self, x, total_training_steps=50000, mask=None, x_cond=None, noise=None, timesteps=None):
        """reference :388-502 signature (plus the two injectable random draws); returns the reference's 7-tuple
        (next_x, current_x, num_timesteps, sigmas, loss_weights, next_noisy_x, current_noisy_x) in NCHW fp32.
        The tensors carry no autograd graph: training goes through forward_nhwc + ops.cm_loss."""
        r = self.forward_nhwc(x, total_training_steps, mask, x_cond, noise, timesteps)
        next_x = ops.cm_combine(r["next_noisy_x"], r["F_next"].detach(), r["cs_n"], r["co_n"])
        current_x = ops.cm_combine(r["current_noisy_x"], r["F_cur"], r["cs_c"], r["co_c"])
        return next_x, current_x, r["num_timesteps"], r["sigmas"], r["loss_weights"], r["next_noisy_x"], r["current_noisy_x"]

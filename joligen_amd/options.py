"""Flat `opt` namespace from a joliGEN JSON config (examples/*.json run unchanged).

The reference assembles `opt` with argparse in three passes (options/base_options.py:102-168,
281-314) and flattens the nested JSON sections with "_" (`G.netG -> G_netG`,
`alg.diffusion.lambda_G -> alg_diffusion_lambda_G`).  The hot path only reads the fields of
SURVEY.md Appendix B; this module reproduces the flattening plus the reference's defaults for
those fields (options/common_options.py, options/train_options.py, models/*_model.py) -- it is
not a re-implementation of the 3000-line option system (CLI help, schema/doc generation).
"""
from __future__ import annotations

import copy
import json
import os
from types import SimpleNamespace

# defaults of the fields the training step reads (reference default in the cited file)
DEFAULTS = dict(
    model_type="palette", name="experiment_name", checkpoints_dir="./checkpoints/", gpu_ids="0", phase="train",
    with_amp=False, with_tf32=False, with_torch_compile=False,
    model_input_nc=3, model_output_nc=3, model_init_type="normal", model_init_gain=0.02, model_multimodal=False,
    model_prior_321_backwardcompatibility=False, model_load_no_strictness=False,
    G_netG="unet_mha", G_ngf=64, G_nblocks=9, G_dropout=False, G_norm="instance", G_spectral=False,
    G_padding_type="reflect", G_diff_n_timestep_train=2000, G_diff_n_timestep_test=1000,
    G_unet_mha_num_heads=1, G_unet_mha_num_head_channels=32, G_unet_mha_res_blocks=[2, 2, 2, 2],
    G_unet_mha_channel_mults=[1, 2, 4, 8], G_unet_mha_attn_res=[16], G_unet_mha_norm_layer="groupnorm",
    G_unet_mha_group_norm_size=32, G_unet_mha_vit_efficient=False,
    alg_palette_loss="MSE", alg_palette_sampling_method="ddpm", alg_palette_minsnr=False,
    alg_diffusion_task="inpainting", alg_diffusion_cond_embed="", alg_diffusion_cond_embed_dim=32,
    alg_diffusion_cond_image_creation="y_t", alg_diffusion_lambda_G=1.0, alg_diffusion_dropout_prob=0.0,
    alg_diffusion_ref_embed_net="clip", alg_diffusion_ddpm_cm_ft=False,
    alg_cm_num_steps=1000000, alg_cm_perceptual_loss=[""], alg_cm_lambda_perceptual=1.0, alg_ddpm_ft_mode="cm", total_iters=0,
    data_crop_size=256, data_load_size=286, data_preprocess="resize_and_crop", data_online_context_pixels=0,
    data_inverted_mask=False, data_refined_mask=False,
    f_s_semantic_nclasses=2, cls_semantic_nclasses=2,
    train_batch_size=1, test_batch_size=1, train_iter_size=1, train_G_ema=False, train_G_ema_beta=0.999,
    train_G_lr=0.0002, train_D_lr=0.0001, train_beta1=0.9, train_beta2=0.999, train_optim="adam",
    train_optim_weight_decay=0.0, train_optim_eps=1e-8, train_pool_size=50, train_continue=False,
    train_continue_from="", train_load_iter=0, train_epoch="latest", train_finetune=False,
    train_lr_policy="linear", train_n_epochs=100, train_n_epochs_decay=100, train_epoch_count=1,
    train_lr_decay_iters=50, train_lr_steps=[], train_feat_wavelet=False, train_metrics_list=[],
    output_display_G_attention_masks=False, output_num_images=20,
    alg_palette_ddim_num_steps=10, alg_palette_ddim_eta=0.5,
    # joligen_amd extensions (not in the reference): activation dtype, fp16 loss scale (0 = default) and how often (in steps) the
    # device-side dropped-step counters are polled to back the scale off (BaseModel.poll_overflow)
    jg_act_dtype="bf16", jg_loss_scale=0.0, jg_overflow_poll=50, jg_async_checkpoint=False, jg_loss_scale_growth_interval=2000,
    jg_projd_backbone="lite0", jg_projd_pretrained="", jg_early_D=True, jg_graph_D=True,
)


def _flatten(d, prefix, out):
    for k, v in d.items():
        key = f"{prefix}_{k}" if prefix else k
        if isinstance(v, dict):
            _flatten(v, key, out)
        else:
            out[key] = v


def opt_from_json(cfg, overrides=None, is_train=True):
    """cfg: path to a reference JSON config or the already-loaded dict.  `overrides` are flat
    (`{"train_batch_size": 32}`), applied last -- like CLI flags after --config_json
    (util/parser.py:33-60)."""
    if isinstance(cfg, str):
        with open(cfg) as f:
            cfg = json.load(f)
    flat = {}
    _flatten(copy.deepcopy(cfg), "", flat)
    vals = dict(DEFAULTS)
    vals.update(flat)
    if overrides:
        vals.update(overrides)
    opt = SimpleNamespace(**vals)
    opt.isTrain = is_train
    # options/common_options.py:1100-1108: "0,1" -> [0, 1]; "-1" -> []
    if isinstance(opt.gpu_ids, str):
        opt.gpu_ids = [int(s) for s in opt.gpu_ids.split(",") if s.strip() != "" and int(s) >= 0]
    # options/train_options.py sanity: G_dropout False -> 0
    opt.G_dropout = float(opt.G_dropout) if not isinstance(opt.G_dropout, bool) else (0.5 if opt.G_dropout else 0.0)
    # options/train_options.py: resuming a run in place and starting from another run's checkpoints exclude each other
    if opt.train_continue and opt.train_continue_from:
        raise ValueError("--train_continue and --train_continue_from are mutually exclusive")
    return opt


def get_train_load_suffix(opt):
    """train.py:92-95: the checkpoint suffix `setup` loads -- `iter_<n>` when train_load_iter > 0, else train_epoch ("latest")"""
    return "iter_%d" % opt.train_load_iter if opt.train_load_iter > 0 else opt.train_epoch


def save_finetune_source_metadata(opt, command_line, model_names):
    """train.py:98-120: with train_continue_from, record where the run's initial weights came from as
    `<checkpoints_dir>/<name>/finetune_source.json` (same keys as the reference writes); no-op otherwise."""
    if not getattr(opt, "train_continue_from", ""):
        return None
    suffix = get_train_load_suffix(opt)
    src = opt.train_continue_from
    meta = {"train_continue_from": src, "train_continue_from_abs": os.path.abspath(os.path.expanduser(src)), "load_suffix": suffix,
            "checkpoint_files": [os.path.join(src, "%s_net_%s.pth" % (suffix, n)) for n in model_names if isinstance(n, str)],
            "command_line": command_line}
    save_dir = os.path.join(opt.checkpoints_dir, opt.name)
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "finetune_source.json")
    with open(path, "w") as f:
        json.dump(meta, f, indent=4)
    return path

"""Generate tests/golden/unet_heads16.pt by running the UNMODIFIED reference (/root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container:   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_heads16.py

`G_unet_mha_num_head_channels = 16` -- the value the reference's own run tests select (tests/test_run_diffusion.py:24) --: the mid-level
attention splits its 64 channels into 4 heads of 16 (models/modules/unet_generator_attn/unet_generator_attn.py:277-347).  UNet forward +
backward on the tiny attention configuration, in the format of oracle/make_golden.py's unet_<cfg>.pt (+ keys / shapes: the weights are
re-derived from them).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
import make_golden as MG  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate
CFG = dict(ngf=32, mults=[1, 2], res_blocks=[1, 1], attn_res=[2], efficient=True, S=16, B=2, num_head_channels=16)


def main():
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    from models import create_model
    from options.train_options import TrainOptions
    import train as ref_train

    c = CFG
    cfg = json.load(open(os.path.join(ref_shim.REFERENCE_ROOT, "examples/example_ddpm_noglasses2glasses.json")))
    cfg["data"]["crop_size"] = cfg["data"]["load_size"] = c["S"]
    cfg["train"]["batch_size"], cfg["train"]["iter_size"] = c["B"], 1
    cfg["gpu_ids"] = "-1"
    cfg["G"].update(ngf=c["ngf"], unet_mha_channel_mults=c["mults"], unet_mha_res_blocks=c["res_blocks"], unet_mha_attn_res=c["attn_res"],
                    unet_mha_vit_efficient=c["efficient"], unet_mha_num_head_channels=c["num_head_channels"])
    cfg["output"]["display"]["type"] = ["none"]
    cfg["checkpoints_dir"], cfg["dataroot"] = "/tmp/jg_golden_ckpt/", "/tmp/nodata"
    opt = TrainOptions().parse_json(cfg, save_config=False)
    opt.use_cuda, opt.optim, opt.jg_dir, opt.total_iters, opt.num_test_images = False, ref_train.optim, ref_shim.REFERENCE_ROOT, 0, 0
    torch.manual_seed(0)
    model = create_model(opt, 0)
    model.setup(opt)
    netG = model.netG_A
    ref_sd = netG.state_dict()
    netG.load_state_dict(O.synth_state_dict(ref_sd, seed=0))
    unet = netG.denoise_fn.model
    heads = [m.num_heads for m in unet.modules() if hasattr(m, "num_heads") and hasattr(m, "qkv")]
    assert heads and all(h == 4 for h in heads), heads          # 64 channels / 16 per head
    g = torch.Generator().manual_seed(12)
    x = torch.randn(c["B"], 6, c["S"], c["S"], generator=g).requires_grad_(True)
    emb = torch.randn(c["B"], 32, generator=g).requires_grad_(True)
    R = torch.randn(c["B"], 3, c["S"], c["S"], generator=g)
    netG.zero_grad()
    out = unet(x, emb)
    (out * R).sum().backward()
    grads = {k: p.grad for k, p in unet.named_parameters()}
    torch.save(dict(cfg=c, x=x.detach(), emb=emb.detach(), R=R, out=out.detach(), dx=x.grad.clone(), demb=emb.grad.clone(), grad_checks=MG.checks(grads),
                    keys=list(ref_sd.keys()), shapes={k: tuple(v.shape) for k, v in ref_sd.items()}),
               os.path.join(OUT, "unet_heads16.pt"))
    print("unet_heads16: heads", heads, "out norm", float(out.norm()))


if __name__ == "__main__":
    main()

"""Step time of the CUT training step (cut_model: resnet_9blocks G + basic PatchGAN D + mlp_sample F, MoNCE, nce_idt, lsgan) on one
MI355X with synthetic images.  Dev tool (GPU box); the judged bench line stays bench.py (palette_model, BASELINE configs[1]).

usage: python tools/cut_bench.py [--size 256] [--batch 4] [--steps 10] [--warmup 3] [--nce monce|patchnce] [--dtype bf16]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from joligen_amd.models import create_model
from joligen_amd.options import opt_from_json

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--nce", default="monce")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--netG", default="resnet", help="resnet | segformer_attn_conv (BASELINE configs[2] generator)")
args = ap.parse_args()

cfg = {"model_type": "cut", "G": {"netG": args.netG, "ngf": 64, "nblocks": 9}, "D": {"netDs": ["basic"], "ndf": 64},
       "alg": {"cut": {"nce_loss": args.nce}}, "data": {"crop_size": args.size, "load_size": args.size},
       "train": {"batch_size": args.batch, "G_ema": False}}
opt = opt_from_json(cfg, overrides={"jg_act_dtype": args.dtype, "gpu_ids": "0"})
model = create_model(opt, 0)
d = torch.device("cuda:0")
g = torch.Generator(device=d).manual_seed(1)
data = {"A": torch.rand(args.batch, 3, args.size, args.size, device=d, generator=g) * 2 - 1,
        "B": torch.rand(args.batch, 3, args.size, args.size, device=d, generator=g) * 2 - 1}
model.data_dependent_initialize(data)


def step():
    model.set_input(data)
    model.optimize_parameters()


for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
losses = {k: round(float(v), 4) for k, v in model.get_current_losses().items()}
print(json.dumps({"workload": f"cut_model {args.netG} G + basic D + mlp_sample F, {args.nce}, {args.size}x{args.size}, batch {args.batch}",
                  "ms_per_step": round(dt * 1e3, 3), "images_per_sec": round(args.batch / dt, 2), "dtype": args.dtype, "losses": losses}))

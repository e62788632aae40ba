"""dev: which python call sites issue aten::copy_ / fill_ / elementwise ATen kernels inside one palette training step (torch.profiler stacks)"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse

import torch
from torch.profiler import ProfilerActivity, profile

import bench

which = sys.argv[1] if len(sys.argv) > 1 else "palette"
if which in ("cut", "mobile", "resnet"):      # BASELINE configs[2] shape; "mobile" / "resnet": the other selections
    args = argparse.Namespace(model="cut", efficient=1, size=256, batch=16, dtype="bf16", netG="segformer_attn_conv", netDs="projected_d,basic", force_exchange=False, proj="vitsmall")
    if which == "mobile":
        args.netG, args.proj = "mobile_resnet_attn", "efficientnet"
    if which == "resnet":
        args.netG, args.netDs, args.proj = "resnet", "basic", "efficientnet"
    os.environ["JG_GRAPH_G"] = os.environ["JG_GRAPH_D"] = "0"      # eager: the profiler sees the ATen ops of the step
    import warnings
    warnings.simplefilter("ignore")
    model, opt = bench.build_model(args, 0, 0, 1)
    g = torch.Generator().manual_seed(77)
    batch = {"A": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).cuda(), "B": (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).cuda()}
else:
    args = argparse.Namespace(model="palette", efficient=1, size=256, batch=32, dtype="bf16", netG="resnet", netDs="basic", force_exchange=False)
    model, opt = bench.build_model(args, 0, 0, 1)
    batch = bench.synth_batch(32, 256, 1234, torch.device("cuda:0"))
for _ in range(3):
    model.set_input(batch)
    model.optimize_parameters()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    model.set_input(batch)
    model.optimize_parameters()
    torch.cuda.synchronize()
cnt = Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::view", "aten::slice", "aten::as_strided", "aten::empty_like", "aten::empty_strided",
                                                          "aten::select", "aten::reshape", "aten::permute", "aten::detach", "aten::alias", "aten::narrow", "aten::_unsafe_view",
                                                          "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::transpose", "aten::t", "aten::to", "aten::resize_",
                                                          "aten::lift_fresh", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::contiguous", "aten::stride"):
        st = [s for s in (ev.stack or []) if "joligen_amd" in s or "bench.py" in s]
        cnt[(ev.name, st[0].strip() if st else "?")] += 1
print("total ATen ops with a device kernel candidate:", sum(cnt.values()))
kern = Counter()
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith("CUDA"):
        kern[ev.name[:60]] += 1
print("device kernels:", sum(kern.values()))
for k, n in kern.most_common(25):
    print(f"   {n:5d} {k}")
for (name, site), n in cnt.most_common(60):
    print(f"{n:5d} {name:28s} {site[:150]}")

shp = Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::add_", "aten::fill_", "aten::zeros", "aten::slice_backward", "aten::cat", "aten::mul", "aten::add", "aten::div", "aten::_to_copy"):
        shp[(ev.name, str(ev.input_shapes)[:110])] += 1
print("by input shapes:")
for (name, sh), n in sorted(shp.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:5d} {name:24s} {sh}")

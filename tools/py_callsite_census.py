"""dev: which python call sites of joligen_amd issue small ATen work (clone / zeros / copy_ / fill_ / add_ / contiguous-copies / cat ...) inside one
CUT training step: the functions are wrapped for one step and the innermost joligen_amd frame of every call is counted"""
import os
import sys
import traceback
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import warnings

import torch

import bench

warnings.simplefilter("ignore")
which = sys.argv[1] if len(sys.argv) > 1 else "cut"
if which == "cut":
    args = argparse.Namespace(model="cut", efficient=1, size=256, batch=16, dtype="bf16", netG="segformer_attn_conv", netDs="projected_d,basic", force_exchange=False)
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
cnt = Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "joligen_amd" in fr.filename and "py_callsite" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def wrap(obj, name, label=None, pred=None):
    orig = getattr(obj, name)

    def w(*a, **k):
        if pred is None or pred(*a, **k):
            cnt[(label or name, site())] += 1
        return orig(*a, **k)

    setattr(obj, name, w)
    return orig


T = torch.Tensor
saved = [(T, n, wrap(T, n)) for n in ("clone", "copy_", "zero_", "fill_", "add_", "mul_", "div_", "float", "sum", "mean", "floor", "__add__", "__mul__", "__truediv__",
                                      "__sub__", "__radd__", "__rmul__", "to")]
saved.append((T, "contiguous", wrap(T, "contiguous", "contiguous(copy)", lambda self, *a, **k: not self.is_contiguous())))
saved += [(torch, n, wrap(torch, n)) for n in ("zeros", "zeros_like", "ones", "cat", "rand", "randperm", "arange", "full", "empty_like", "stack", "where")]
model.set_input(batch)
model.optimize_parameters()
torch.cuda.synchronize()
for o, n, f in saved:
    setattr(o, n, f)
print("wrapped python-level calls in one step:", sum(cnt.values()))
for (name, s), n in cnt.most_common(70):
    print(f"{n:5d} {name:18s} {s}")

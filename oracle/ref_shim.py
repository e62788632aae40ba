"""Import shim for running the UNMODIFIED reference (/root/reference) in this container.

TEST INFRASTRUCTURE ONLY (oracle/): used by oracle/make_golden.py to generate the
fixtures under tests/golden/.  Nothing in the product path imports this file, and
nothing here runs on the GPU box (/root/reference does not exist there).

The reference's `models/__init__.py -> base_model.py` imports a long list of third-party
modules at import time that are absent here.  We insert module stubs with
a valid `__spec__` (a bare MagicMock in sys.modules breaks torch._dynamo's find_spec)
whose attributes resolve to MagicMock.  Only arithmetic implemented by torch itself is
ever executed through this shim.
"""
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get("JG_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "cv2", "thop", "torchviz", "piq", "lpips",
    "torchvision", "torchvision.transforms", "torchvision.transforms.v2",
    "torchvision.transforms.functional", "torchvision.transforms.v2.functional",
    "torchvision.ops", "torchvision.models", "torchvision.utils", "torchvision.io",
    "torchvision.models.feature_extraction", "torchvision.transforms.functional_tensor",
    "positional_encodings", "positional_encodings.torch_encodings",
    "clip", "bitsandbytes", "visdom", "wget", "dominate", "dominate.tags",
    "timm", "timm.models", "timm.models.layers", "timm.layers", "timm.models.vision_transformer",
    "segment_anything", "segment_anything.modeling", "segment_anything.modeling.common",
    "segment_anything.modeling.image_encoder", "segment_anything.modeling.mask_decoder",
    "segment_anything.modeling.prompt_encoder", "segment_anything.modeling.transformer",
    "segment_anything.modeling.sam", "segment_anything.utils", "segment_anything.utils.transforms",
    "segment_anything.utils.amg",
    "mobile_sam", "mobile_sam.modeling", "mobile_sam.utils", "mobile_sam.utils.transforms",
    "aim", "imgaug", "imgaug.augmenters", "diffusers", "diffusers.utils",
    "diffusers.utils.peft_utils", "diffusers.models", "peft", "ftfy", "kornia",
    "kornia.filters", "skimage", "skimage.metrics", "matplotlib", "matplotlib.pyplot",
    "ot", "torch_dct", "h5py", "torchinfo", "pycocotools", "pycocotools.mask",
]


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        m = MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, m)
        return m


def install():
    """Insert stubs for missing third-party modules and put the reference on sys.path."""
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            if importlib.util.find_spec(name.split(".")[0]) is not None and "." not in name:
                # real module exists; keep it
                __import__(name)
                continue
        except Exception:
            pass
        if name.split(".")[0] in sys.modules and not isinstance(sys.modules[name.split(".")[0]], _Stub):
            continue
        mod = _Stub(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=True)
        mod.__path__ = []
        mod.__file__ = f"<stub {name}>"
        sys.modules[name] = mod
        if "." in name:
            parent, child = name.rsplit(".", 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], child, mod)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    patch_instance_norm_backward()


def patch_instance_norm_backward():
    """Work around a PyTorch CPU bug (seen with torch 2.10): the backward of `torch.nn.functional.instance_norm` returns WRONG
    values when grad_output has channels-last strides and the input is contiguous (finite differences side with the contiguous
    path; tests/test_oracle_golden.py::test_torch_cpu_instance_norm_channels_last_backward pins this).  The reference hits it on
    CPU only: PatchSampleF's `feat.permute(0, 2, 3, 1).flatten(1, 2)` (cut_networks.py:45) sends a channels-last gradient into the
    `x + InstanceNorm(...)` of the tapped ResnetBlocks.  GPU runs of the reference do not have the bug, so the fixtures are
    generated with grad_output made contiguous -- the reference's code is untouched, only the framework function is wrapped."""
    import torch
    import torch.nn.functional as F

    if getattr(F.instance_norm, "_jg_contig_grad", False):
        return
    orig = F.instance_norm

    class _ContigGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()      # a fresh tensor: the reference's in-place nn.ReLU(True) then modifies this output

        @staticmethod
        def backward(ctx, g):
            return g.contiguous()

    def instance_norm(*args, **kwargs):
        return _ContigGrad.apply(orig(*args, **kwargs))

    instance_norm._jg_contig_grad = True
    instance_norm._jg_orig = orig
    F.instance_norm = instance_norm


import importlib.util  # noqa: E402

"""Generate tests/golden/cutnet_*.pt: forward / get_feats / backward of the UNMODIFIED reference ResnetGenerator and
NLayerDiscriminator modules on CPU (TEST INFRASTRUCTURE ONLY).
   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_cut.py"""
import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import jg_oracle as O  # noqa: E402
from make_golden import checks  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate
CFGS = {"small": dict(ngf=16, n_blocks=2, ndf=16, S=32, B=2), "wide": dict(ngf=64, n_blocks=3, ndf=64, S=32, B=1)}
NCE_LAYERS = [0, 4, 8, 10, 11]


def main():
    os.makedirs(OUT, exist_ok=True)
    from models.modules.discriminators import NLayerDiscriminator
    from models.modules.resnet_architecture.resnet_generator import ResnetGenerator

    norm = functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)   # models/modules/utils.py:101-104
    for name, c in CFGS.items():
        g = torch.Generator().manual_seed(5)
        x = torch.rand(c["B"], 3, c["S"], c["S"], generator=g) * 2 - 1
        netG = ResnetGenerator(3, 3, c["ngf"], norm_layer=norm, use_dropout=False, n_blocks=c["n_blocks"], padding_type="reflect")
        sdG = O.synth_state_dict(netG.state_dict(), seed=0)
        netG.load_state_dict(sdG)
        xg = x.clone().requires_grad_(True)
        out = netG(xg)
        R = torch.randn(out.shape, generator=g)
        (out * R).sum().backward()
        feats = netG.get_feats(x, list(NCE_LAYERS))
        recG = dict(x=x, R=R, out=out.detach(), dx=xg.grad.clone(), feats=[f.detach() for f in feats],
                    grad_checks=checks({k: p.grad for k, p in netG.named_parameters()}), keys=list(sdG.keys()),
                    shapes={k: tuple(v.shape) for k, v in sdG.items()})
        netD = NLayerDiscriminator(3, c["ndf"], n_layers=3, norm_layer=norm, use_dropout=False, use_spectral=False)
        sdD = O.synth_state_dict(netD.state_dict(), seed=1)
        netD.load_state_dict(sdD)
        xd = x.clone().requires_grad_(True)
        pred = netD(xd)
        Rd = torch.randn(pred.shape, generator=g)
        (pred * Rd).sum().backward()
        recD = dict(R=Rd, out=pred.detach(), dx=xd.grad.clone(), grad_checks=checks({k: p.grad for k, p in netD.named_parameters()}),
                    keys=list(sdD.keys()), shapes={k: tuple(v.shape) for k, v in sdD.items()})
        torch.save(dict(cfg=c, nce_layers=NCE_LAYERS, G=recG, D=recD), os.path.join(OUT, f"cutnet_{name}.pt"))
        print(name, tuple(out.shape), tuple(pred.shape), [tuple(f.shape) for f in feats])


if __name__ == "__main__":
    main()

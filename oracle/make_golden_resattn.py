"""Generate tests/golden/resattn_{plain,mobile}.pt: forward / get_feats / backward of the UNMODIFIED reference
ResnetGenerator_attn (G_netG = resnet_attn / mobile_resnet_attn) on CPU (TEST INFRASTRUCTURE ONLY).
   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_resattn.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402
from make_golden import checks  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate
CFG = dict(ngf=16, n_blocks=3, S=64, B=2, nb_mask_attn=10, nb_mask_input=1)
NCE_LAYERS = [0, 2, 4, 8]      # ids 4 and 8 are beyond the 3 blocks: they tap nothing (compute_feats :504-515)


def main():
    os.makedirs(OUT, exist_ok=True)
    from models.modules.resnet_architecture.resnet_generator import ResnetGenerator_attn

    c = CFG
    for name, mobile in (("plain", False), ("mobile", True)):
        g = torch.Generator().manual_seed(11)
        x = torch.rand(c["B"], 3, c["S"], c["S"], generator=g) * 2 - 1
        net = ResnetGenerator_attn(3, 3, c["nb_mask_attn"], c["nb_mask_input"], c["ngf"], n_blocks=c["n_blocks"], use_spectral=False,
                                   padding_type="reflect", mobile=mobile)
        sd = O.synth_state_dict(net.state_dict(), seed=0)
        net.load_state_dict(sd)
        xg = x.clone().requires_grad_(True)
        out = net(xg)
        R = torch.randn(out.shape, generator=g)
        (out * R).sum().backward()
        feats = net.get_feats(x, list(NCE_LAYERS))
        rec = dict(cfg=c, mobile=mobile, nce_layers=NCE_LAYERS, x=x, R=R, out=out.detach(), dx=xg.grad.clone(),
                   feats=[f.detach() for f in feats], grad_checks=checks({k: p.grad for k, p in net.named_parameters()}),
                   keys=list(sd.keys()), shapes={k: tuple(v.shape) for k, v in sd.items()})
        torch.save(rec, os.path.join(OUT, f"resattn_{name}.pt"))
        print(name, tuple(out.shape), [tuple(f.shape) for f in feats], len(sd))


if __name__ == "__main__":
    main()

"""Generate tests/golden/segformer_*.pt: forward / get_feats / backward of the UNMODIFIED reference SegformerGenerator_attn
(G_netG = segformer_attn_conv: MiT-b0 from scratch + ResnetDecoder tail) on CPU in TRAIN mode, with the DropPath / Dropout2d
uniforms recorded.  TEST INFRASTRUCTURE ONLY.
   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_segformer.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import jg_oracle as O  # noqa: E402
from make_golden import checks  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate
CFGS = {"s64": dict(S=64, B=2), "s128": dict(S=128, B=1)}


class Recorder:
    """Replaces torch.rand (DropPath) and F.dropout2d (heads) by equivalents that log the uniforms they consume."""

    def __init__(self, seed):
        self.g, self.log = torch.Generator().manual_seed(seed), []
        self.real_rand, self.real_d2 = torch.rand, F.dropout2d

    def rand(self, *shape, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        u = self.real_rand(tuple(shape), generator=self.g)
        self.log.append(u.clone())
        return u.to(kw.get("dtype") or torch.float32)

    def dropout2d(self, input, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return input
        u = self.real_rand((input.shape[0], input.shape[1]), generator=self.g)
        self.log.append(u.clone())
        return input * (u >= p).to(input.dtype).view(input.shape[0], input.shape[1], 1, 1) / (1 - p)

    def __enter__(self):
        torch.rand, F.dropout2d = self.rand, self.dropout2d
        return self

    def __exit__(self, *a):
        torch.rand, F.dropout2d = self.real_rand, self.real_d2


def main():
    os.makedirs(OUT, exist_ok=True)
    from models.modules.segformer.segformer_generator import SegformerGenerator_attn

    for name, c in CFGS.items():
        torch.manual_seed(0)
        net = SegformerGenerator_attn(ref_shim.REFERENCE_ROOT, "models/configs/segformer/segformer_config_b0.json", 3, img_size=c["S"],
                                      nb_mask_attn=10, nb_mask_input=1, final_conv=True, padding_type="reflect")
        net.train()
        sd = O.synth_state_dict(net.state_dict(), seed=4)
        net.load_state_dict(sd)
        g = torch.Generator().manual_seed(8)
        x = torch.rand(c["B"], 3, c["S"], c["S"], generator=g) * 2 - 1
        xg = x.clone().requires_grad_(True)
        with Recorder(17) as rec:
            out = net(xg)
            n_fwd = len(rec.log)
            R = torch.randn(out.shape, generator=g)
            (out * R).sum().backward()
            grads = {k: p.grad.clone() for k, p in net.named_parameters()}
            dx = xg.grad.clone()
            feats = net.get_feats(x, [0, 1, 2, 3])
            rands = [u.clone() for u in rec.log]
        bn_after = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
        net.eval()
        with torch.no_grad():
            out_eval = net(x)
        torch.save(dict(cfg=c, x=x, R=R, out=out.detach(), out_eval=out_eval, dx=dx, feats=[f.detach() for f in feats], rands=rands, n_fwd=n_fwd, bn_after=bn_after,
                        grad_checks=checks(grads), keys=list(sd.keys()), shapes={k: tuple(v.shape) for k, v in sd.items()}),
                   os.path.join(OUT, f"segformer_{name}.pt"))
        print(name, tuple(out.shape), [tuple(f.shape) for f in feats], "uniform draws:", [tuple(u.shape) for u in rands[:4]], len(rands), n_fwd,
              "params", sum(v.numel() for v in sd.values()))
    print({f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT)) if f.startswith("segformer")})


if __name__ == "__main__":
    main()

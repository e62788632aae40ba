"""Generate tests/golden/palette_loss.pt from the UNMODIFIED reference on CPU (TEST INFRASTRUCTURE ONLY): torch.nn.L1Loss and
MultiScaleDiffusionLoss (models/modules/loss.py:397-467) called exactly as PaletteModel.compute_palette_loss does
(palette_model.py:597-618), values and gradients with respect to noise_hat.
   PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_palette_loss.py"""
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402

import jg_oracle as O  # noqa: E402

OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")   # JG_GOLDEN_OUT: tests/test_oracle_golden.py::test_fixtures_regenerate


def main():
    from models.modules.loss import MultiScaleDiffusionLoss

    recs = {}
    for S, B in ((128, 2), (64, 1), (32, 2)):
        inp = O.palette_loss_inputs(S, B)
        noise, mask, w = inp["noise"], inp["mask"], inp["w"]
        noise_hat = inp["noise_hat"].clone().requires_grad_(True)
        scales = [2 ** k for k in range(5, math.floor(math.log2(S)) + 1)]      # palette_model.py:232-241
        for lossname in ("L1", "multiscale_L1", "multiscale_MSE"):
            fn = torch.nn.L1Loss() if lossname == "L1" else MultiScaleDiffusionLoss(lossname, img_size=S, scales=scales)
            for use_mask in (True, False):
                for use_w in (False, True):
                    ww = w if use_w else 1.0
                    if use_mask:
                        mb = torch.clamp(mask, min=0, max=1)
                        loss = fn(ww * mb * noise, ww * mb * noise_hat)
                    else:
                        loss = fn(ww * noise, ww * noise_hat)
                    levels = {}
                    if isinstance(loss, dict):
                        levels = {k: v.detach().clone() for k, v in loss.items()}
                        loss = sum(loss.values())
                    (gr,) = torch.autograd.grad(loss, noise_hat)
                    rec = dict(loss=loss.detach().clone(), levels=levels,
                               grad_check=torch.stack([gr.norm(), (gr * O.projection_vector("palette_loss_grad", gr.shape)).sum()]))
                    if S == 32:
                        rec["grad"] = gr.clone()          # the full gradient only at the smallest size (fixture size)
                    recs[(S, lossname, use_mask, use_w)] = rec
        recs[("inputs", S)] = dict(B=B, check=float(noise_hat.detach().double().sum()))
    torch.save(recs, os.path.join(OUT, "palette_loss.pt"))
    print({k: (float(v["loss"]), sorted(v["levels"])) for k, v in recs.items() if k[0] != "inputs" and k[2] and not k[3]})
    print("bytes", os.path.getsize(os.path.join(OUT, "palette_loss.pt")))


if __name__ == "__main__":
    main()

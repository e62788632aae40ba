"""tests/golden/resize_pil.pt: PIL `Image.resize` outputs (BICUBIC for images, NEAREST for masks) on seeded uint8 inputs -- what torchvision's
`transforms.Resize` / the reference's ResizeMask (data/base_dataset.py:441-443,749-763) compute on PIL images.  TEST INFRASTRUCTURE ONLY.
   python oracle/make_golden_resize.py      (needs Pillow; no /root/reference import: the transform IS the Pillow call)"""
import os

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("JG_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")
CASES = [(40, 56, 32, 32), (37, 29, 48, 48), (64, 64, 24, 24), (30, 45, 30, 60), (96, 80, 71, 33)]      # (H, W, out_h, out_w): down, up, mixed, odd


def main():
    g = torch.Generator().manual_seed(12)
    out = []
    for H, W, oh, ow in CASES:
        img = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
        # a smooth component next to the noise: resampling of real images is not all high frequency
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        img = ((img.float() * 0.5) + 127 * (1 + torch.sin(yy / 7.0 + xx / 5.0))[..., None] * 0.5).clamp(0, 255).to(torch.uint8)
        mask = (torch.rand(H, W, generator=g) < 0.4).to(torch.uint8) * torch.randint(1, 5, (H, W), generator=g, dtype=torch.uint8)
        r_img = np.asarray(Image.fromarray(img.numpy()).resize((ow, oh), Image.BICUBIC))
        r_mask = np.asarray(Image.fromarray(mask.numpy()).resize((ow, oh), Image.NEAREST))
        out.append(dict(img=img, mask=mask, out_hw=(oh, ow), img_resized=torch.from_numpy(r_img.copy()), mask_resized=torch.from_numpy(r_mask.copy())))
    import PIL

    torch.save(dict(cases=out, pillow=PIL.__version__), os.path.join(OUT, "resize_pil.pt"))
    print("resize_pil.pt", len(out), "cases, Pillow", PIL.__version__)


if __name__ == "__main__":
    main()

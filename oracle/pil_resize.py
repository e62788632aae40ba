"""CPU restatement of the reference's input transforms for the device input pipeline -- TEST INFRASTRUCTURE ONLY.

The reference resizes with torchvision (`transforms.Resize(osize, interpolation=BICUBIC)`, /root/reference/data/base_dataset.py:441-443;
masks `InterpolationMode.NEAREST`, :749-763) on PIL images, i.e. with Pillow's `Image.resize`.  torchvision (pinned 0.19.0 in
requirements.txt) is ABSENT here; Pillow (unpinned in requirements.txt, 12.2.0 in this image) is present and is the library that does the
arithmetic, so the restatement below follows Pillow's published algorithm (src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; Geometry.c nearest filter) and is PINNED against Pillow itself:
tests/golden/resize_pil.pt is written by oracle/make_golden_resize.py from `Image.resize` outputs.
Then ToTensor / Normalize / crop / flip / fill_mask_with_random in torchvision's order of operations (base_dataset.py:513-528,
data/online_creation.py:1366-1376)."""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support=2.0):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box: (bounds int32 [out, 2], kk int32 [out, ksize])"""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """one separable pass over a uint8 [H, W, C] array (axis 1 = horizontal, 0 = vertical)"""
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.zeros((bounds.shape[0],) + src.shape[1:], np.int64)
    for o in range(bounds.shape[0]):
        lo, n = bounds[o]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for j in range(n):
            acc += src[lo + j] * int(kk[o, j])
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_bicubic_u8(img, out_h, out_w):
    """Image.resize((out_w, out_h), BICUBIC) of a uint8 [H, W, 3] array: horizontal pass first, then vertical (ImagingResample)"""
    H, W, _ = img.shape
    if out_w != W:
        img = _pass(img, *precompute_coeffs(W, out_w), axis=1)
    if out_h != H:
        img = _pass(img, *precompute_coeffs(H, out_h), axis=0)
    return img


def nearest_index_table(in_size, out_size):
    """Geometry.c ImagingScaleAffine: the source index of every output position, with PIL's INCREMENTAL double accumulation
    (xo = a * 0.5; xin = (int) xo; xo += a) -- not floor((x + 0.5) * a): the two differ where (x + 0.5) * a is an integer"""
    a = in_size / out_size
    xo = a * 0.5
    tab = np.zeros(out_size, np.int64)
    for x in range(out_size):
        xin = int(xo) if xo >= 0.0 else -1
        tab[x] = min(max(xin, 0), in_size - 1)
        xo += a
    return tab


def resize_nearest_u8(mask, out_h, out_w):
    """Image.resize((out_w, out_h), NEAREST) of a uint8 [H, W] array"""
    H, W = mask.shape
    return mask[nearest_index_table(H, out_h)][:, nearest_index_table(W, out_w)]


def input_pipeline_reference(img_u8, mask_u8, offsets, flips, noise, crop, load_size=None):
    """the reference's per-image transform chain on a batch: [Resize(load_size)] -> ToTensor (uint8 / 255) -> Normalize((x - 0.5) / 0.5)
    -> crop window -> horizontal flip -> fill_mask_with_random (A = B * (1 - m) + noise * m, m = mask != 0).
    img_u8 [B, H, W, 3] uint8, mask_u8 [B, H, W] uint8, offsets [B, 2] (oy, ox), flips [B] bool, noise [B, 3, S, S].  Returns A, B, mask."""
    B = img_u8.shape[0]
    S = crop
    As, Bs, Ms = [], [], []
    for b in range(B):
        im, mk = img_u8[b].numpy(), mask_u8[b].numpy()
        if load_size is not None:
            im, mk = resize_bicubic_u8(im, load_size, load_size), resize_nearest_u8(mk, load_size, load_size)
        x = torch.from_numpy(im).permute(2, 0, 1).float().div(255).sub(0.5).div(0.5)
        m = torch.from_numpy(mk)
        oy, ox = int(offsets[b][0]), int(offsets[b][1])
        xb, mb = x[:, oy:oy + S, ox:ox + S], m[oy:oy + S, ox:ox + S]
        if bool(flips[b]):
            xb, mb = xb.flip(-1), mb.flip(-1)
        m01 = torch.where(mb != 0, 1.0, 0.0)[None]
        As.append(xb * (1 - m01) + noise[b] * m01)
        Bs.append(xb)
        Ms.append(mb[None].long())
    return torch.stack(As), torch.stack(Bs), torch.stack(Ms)

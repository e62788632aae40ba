"""Device-side input pipeline: crop / flip / normalise / mask fill on the GPU, H2D staging through pinned buffers on a
copy stream, so that an 600+ images/s training step is not starved by CPU DataLoader workers (the reference does all of this per
image in `data/__init__.py:138-146` worker processes: `data/base_dataset.py:513-528,892-1003`, `data/online_creation.py:1366-1376`,
`data/self_supervised_labeled_mask_dataset.py:46-62`).

The host side only has to hand over DECODED uint8 images (HWC) and uint8 label masks plus a crop window per image; one fused kernel
(`jg_input_pipeline`) produces the batch dict `PaletteModel.set_input` / `CMModel.set_input` consume: {"A", "B", "B_label_mask"}.
"""
from __future__ import annotations

import math

import torch

from . import _lib
from ._lib import check


def pil_bicubic_tables(in_size, out_size):
    """fixed-point tap tables of Pillow's BICUBIC resampler for one axis (Resample.c precompute_coeffs + normalize_coeffs_8bpc, full-image
    box): bounds int32 [out, 2] = (first source index, tap count), kk int32 [out, ksize] = taps with 22 fractional bits.  Computed on the
    host in double precision exactly as Pillow does -- the device passes (jg_resample_u8) are then integer arithmetic, bit-exact with
    `transforms.Resize(osize, BICUBIC)` on a PIL image (reference data/base_dataset.py:441-443)."""
    def bicubic(x, a=-0.5):
        x = abs(x)
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0

    scale = in_size / out_size
    fscale = max(scale, 1.0)
    sup = 2.0 * fscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = torch.zeros((out_size, 2), dtype=torch.int32)
    kk = torch.zeros((out_size, ksize), dtype=torch.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        n = min(int(center + sup + 0.5), in_size) - xmin
        w = [bicubic((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = sum(w)
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx, 0], bounds[xx, 1] = xmin, n
    return bounds, kk


def pil_nearest_table(in_size, out_size):
    """source index of every output position of Pillow's NEAREST resize (Geometry.c ImagingScaleAffine: the source coordinate is
    accumulated incrementally in double, xo = a / 2; xo += a) -- the label-mask resize of the reference's ResizeMask (:749-763)"""
    a = in_size / out_size
    xo = a * 0.5
    tab = torch.zeros(out_size, dtype=torch.int32)
    for x in range(out_size):
        tab[x] = min(max(int(xo) if xo >= 0.0 else -1, 0), in_size - 1)
        xo += a
    return tab


class DeviceInputPipeline:
    """`submit()` stages one batch (async H2D on a private copy stream into one of `n_buffers` pinned / device slots); `get()` returns
    the oldest staged batch as device tensors, the consumer's stream waiting (on the device) for the copy + kernel."""

    def __init__(self, crop_size, device, n_buffers=2, load_size=None):
        """load_size: the reference's `data_load_size` with data_preprocess = "resize_and_crop" (base_dataset.py:441-443): every image is
        first resized to load_size x load_size (PIL BICUBIC; the label mask NEAREST) on the device, then cropped to crop_size"""
        self.S = int(crop_size)
        self.load_size = None if load_size is None else int(load_size)
        self._tables = {}
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.n_buffers = n_buffers
        self._slots = [None] * n_buffers      # per slot: dict of pinned host + device staging tensors (allocated at first use)
        self._queue = []
        self._next = 0

    def _slot(self, B, H, W, with_mask):
        i = self._next
        self._next = (self._next + 1) % self.n_buffers
        s = self._slots[i]
        if s is None or s["shape"] != (B, H, W, with_mask):
            S = self.S
            s = dict(shape=(B, H, W, with_mask),
                     h_img=torch.empty((B, H, W, 3), dtype=torch.uint8, pin_memory=True),
                     h_mask=torch.empty((B, H, W), dtype=torch.uint8, pin_memory=True) if with_mask else None,
                     h_win=torch.empty((B, 3), dtype=torch.int32, pin_memory=True),
                     d_img=torch.empty((B, H, W, 3), dtype=torch.uint8, device=self.device),
                     d_mask=torch.empty((B, H, W), dtype=torch.uint8, device=self.device) if with_mask else None,
                     d_win=torch.empty((B, 3), dtype=torch.int32, device=self.device),
                     event=None)
            self._slots[i] = s
        elif s["event"] is not None:
            s["event"].synchronize()         # the slot's previous batch has left the staging buffers
        return s

    def submit(self, imgs_u8, masks_u8, offsets, flips=None, noise=None, paths=None):
        """imgs_u8: uint8 [B,H,W,3] (CPU); masks_u8: uint8 [B,H,W] or None; offsets: [B,2] (oy, ox) crop origins; flips: [B] bool;
        noise: optional fp32 [B,3,S,S] N(0,1) draws (parity runs), else drawn on the device."""
        B, H, W, _ = imgs_u8.shape
        S = self.S
        s = self._slot(B, H, W, masks_u8 is not None)
        s["h_img"].copy_(imgs_u8)
        if masks_u8 is not None:
            s["h_mask"].copy_(masks_u8)
        s["h_win"][:, :2] = torch.as_tensor(offsets, dtype=torch.int32)
        s["h_win"][:, 2] = 0 if flips is None else torch.as_tensor(flips, dtype=torch.int32)
        L = self.load_size
        Hc, Wc = (L, L) if L is not None else (H, W)          # the size the crop windows refer to
        if int(s["h_win"][:, 0].max()) + S > Hc or int(s["h_win"][:, 1].max()) + S > Wc or int(s["h_win"][:, :2].min()) < 0:
            raise ValueError("crop window outside the source image")
        with torch.cuda.stream(self.stream):
            s["d_img"].copy_(s["h_img"], non_blocking=True)
            if masks_u8 is not None:
                s["d_mask"].copy_(s["h_mask"], non_blocking=True)
            s["d_win"].copy_(s["h_win"], non_blocking=True)
            d_img, d_mask = s["d_img"], s["d_mask"]
            if L is not None and (H, W) != (L, L):
                d_img, d_mask = self._resize(d_img, d_mask if masks_u8 is not None else None, B, H, W, L)
                H, W = L, L
            A = torch.empty((B, 3, S, S), device=self.device, dtype=torch.float32)
            Bimg = torch.empty_like(A)
            m = torch.empty((B, 1, S, S), device=self.device, dtype=torch.int64)
            nz = noise.to(self.device, non_blocking=True).float().contiguous() if noise is not None else \
                (torch.randn((B, 3, S, S), device=self.device) if masks_u8 is not None else None)
            check(_lib.lib().jg_input_pipeline(d_img.data_ptr(), None if masks_u8 is None else d_mask.data_ptr(), s["d_win"].data_ptr(),
                                               None if nz is None else nz.data_ptr(), A.data_ptr(), Bimg.data_ptr(), m.data_ptr(), B, H, W, S,
                                               self.stream.cuda_stream), "jg_input_pipeline")
            ev = torch.cuda.Event()
            ev.record(self.stream)
        s["event"] = ev
        self._queue.append((ev, {"A": A, "B": Bimg, "B_label_mask": m, "A_img_paths": paths or ["device"] * B}))

    def _resize(self, d_img, d_mask, B, H, W, L):
        """[B, H, W, 3] uint8 -> [B, L, L, 3] (two fixed-point passes, horizontal first like PIL's ImagingResample), mask nearest"""
        key = (H, W, L)
        if key not in self._tables:
            bw, kw = pil_bicubic_tables(W, L)
            bh, kh = pil_bicubic_tables(H, L)
            self._tables[key] = tuple(t.to(self.device) for t in (bw, kw, bh, kh, pil_nearest_table(H, L), pil_nearest_table(W, L)))
        bw, kw, bh, kh, ty, tx = self._tables[key]
        st = self.stream.cuda_stream
        L_ = _lib.lib()
        tmp = torch.empty((B, H, L, 3), dtype=torch.uint8, device=self.device)
        out = torch.empty((B, L, L, 3), dtype=torch.uint8, device=self.device)
        check(L_.jg_resample_u8(d_img.data_ptr(), tmp.data_ptr(), bw.data_ptr(), kw.data_ptr(), kw.shape[1], 0, B, H, W, H, L, st), "jg_resample_u8")
        check(L_.jg_resample_u8(tmp.data_ptr(), out.data_ptr(), bh.data_ptr(), kh.data_ptr(), kh.shape[1], 1, B, H, L, L, L, st), "jg_resample_u8")
        m = None
        if d_mask is not None:
            m = torch.empty((B, L, L), dtype=torch.uint8, device=self.device)
            check(L_.jg_resize_nearest_u8(d_mask.data_ptr(), m.data_ptr(), ty.data_ptr(), tx.data_ptr(), B, H, W, L, L, st), "jg_resize_nearest_u8")
        return out, m

    def get(self):
        ev, batch = self._queue.pop(0)
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(cur)       # allocated on the copy stream, consumed on the compute stream
        return batch

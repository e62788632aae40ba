"""Device-side input pipeline: crop / flip / normalise / mask fill on the GPU, H2D staging through pinned buffers on a
copy stream, so that an 600+ images/s training step is not starved by CPU DataLoader workers (the reference does all of this per
image in `data/__init__.py:138-146` worker processes: `data/base_dataset.py:513-528,892-1003`, `data/online_creation.py:1366-1376`,
`data/self_supervised_labeled_mask_dataset.py:46-62`).

The host side only has to hand over DECODED uint8 images (HWC) and uint8 label masks plus a crop window per image; one fused kernel
(`jg_input_pipeline`) produces the batch dict `PaletteModel.set_input` / `CMModel.set_input` consume: {"A", "B", "B_label_mask"}.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check


class DeviceInputPipeline:
    """`submit()` stages one batch (async H2D on a private copy stream into one of `n_buffers` pinned / device slots); `get()` returns
    the oldest staged batch as device tensors, the consumer's stream waiting (on the device) for the copy + kernel."""

    def __init__(self, crop_size, device, n_buffers=2):
        self.S = int(crop_size)
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
        if int(s["h_win"][:, 0].max()) + S > H or int(s["h_win"][:, 1].max()) + S > W or int(s["h_win"][:, :2].min()) < 0:
            raise ValueError("crop window outside the source image")
        with torch.cuda.stream(self.stream):
            s["d_img"].copy_(s["h_img"], non_blocking=True)
            if masks_u8 is not None:
                s["d_mask"].copy_(s["h_mask"], non_blocking=True)
            s["d_win"].copy_(s["h_win"], non_blocking=True)
            A = torch.empty((B, 3, S, S), device=self.device, dtype=torch.float32)
            Bimg = torch.empty_like(A)
            m = torch.empty((B, 1, S, S), device=self.device, dtype=torch.int64)
            nz = noise.to(self.device, non_blocking=True).float().contiguous() if noise is not None else \
                (torch.randn((B, 3, S, S), device=self.device) if masks_u8 is not None else None)
            check(_lib.lib().jg_input_pipeline(s["d_img"].data_ptr(), None if masks_u8 is None else s["d_mask"].data_ptr(), s["d_win"].data_ptr(),
                                               None if nz is None else nz.data_ptr(), A.data_ptr(), Bimg.data_ptr(), m.data_ptr(), B, H, W, S,
                                               self.stream.cuda_stream), "jg_input_pipeline")
            ev = torch.cuda.Event()
            ev.record(self.stream)
        s["event"] = ev
        self._queue.append((ev, {"A": A, "B": Bimg, "B_label_mask": m, "A_img_paths": paths or ["device"] * B}))

    def get(self):
        ev, batch = self._queue.pop(0)
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(cur)       # allocated on the copy stream, consumed on the compute stream
        return batch

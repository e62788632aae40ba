"""History buffer of generated images for the discriminator update: behaviour of /root/reference/util/image_pool.py
(`ImagePool.query` :19-56, `get_random` :64-72).  The host RNG is an attribute (`rng`, default: the python `random` module,
as in the reference) so that parity runs can inject the draws."""
from __future__ import annotations

import random

import torch


class ImagePool:
    def __init__(self, pool_size, rng=None):
        self.pool_size = pool_size
        self.rng = rng if rng is not None else random
        if self.pool_size > 0:
            self.num_imgs = 0
            self.images = []

    def query(self, images, reader_stream=None):
        """Each incoming image: stored and returned while the pool fills; afterwards with probability 1/2 swapped against a
        random stored image (which is returned instead), else returned as is.  Draw order: uniform(0,1) then randint.
        reader_stream: the (non-allocating) stream this call runs on when that is not the stream the stored images were allocated on
        (cut_model's discriminator stream): a stored image is marked as read there before its last reference is dropped, so the caching
        allocator does not hand its block out again while the copy is still queued."""
        if self.pool_size == 0:
            return images
        out = []
        for image in images:
            image = image.detach().unsqueeze(0)
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image)
                out.append(image)
            elif self.rng.uniform(0, 1) > 0.5:
                idx = self.rng.randint(0, self.pool_size - 1)
                if reader_stream is not None:
                    self.images[idx].record_stream(reader_stream)
                old = self.images[idx].clone()
                self.images[idx] = image
                out.append(old)
            else:
                out.append(image)
        return torch.cat(out, 0)

    def store(self, images):
        """`query` for a caller that discards the result (cut_model.forward's real-image pools, base_gan_model.py:172-173 of the reference): the same
        pool update from the same host draws, without the device copies of the returned batch (round 6: ~18 launches per step)."""
        if self.pool_size == 0:
            return
        for image in images:
            image = image.detach().unsqueeze(0)
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image)
            elif self.rng.uniform(0, 1) > 0.5:
                self.images[self.rng.randint(0, self.pool_size - 1)] = image

    def get_all(self):
        return self.images

    def __len__(self):
        return len(self.images)

    def get_random(self, nb):
        return torch.cat([self.images[self.rng.randint(0, len(self.images) - 1)].clone() for _ in range(nb)], 0)

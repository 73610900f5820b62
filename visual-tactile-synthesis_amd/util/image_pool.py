"""History buffer of generated images for the discriminator update (reference: util/image_pool.py:10-61; pix2pixHD's `fake_pool`,
models/pix2pixHD_model.py:334, 582).

The reference walks the batch image by image: while the pool is not full the image is stored and returned; afterwards
`random.uniform(0, 1) > 0.5` swaps it with the slot `random.randint(0, pool_size - 1)` (the slot's old content is returned), else the image
itself is returned.  Here the DECISIONS are made on the host with the same calls to Python's `random` in the same order (`plan`), and the
data movement is one device kernel over all images (`vts_pool_query`) that a captured step replays: the plan travels as two small int32
arrays refreshed before every step.  Several tensors that the reference pools as one concatenation (label ++ image) are pooled as parallel
stores under one plan.
"""
import random

import torch

from vts import ops


class ImagePool:
    def __init__(self, pool_size):
        self.pool_size = int(pool_size)
        self.num_imgs = 0
        self._stores = {}        # name -> [pool_size, C, H, W]
        self._slots = None       # device int32 [2, n]: row 0 = slot returned instead of image n (-1: itself), row 1 = slot image n goes to
        self._host = None

    def plan(self, n):
        """decisions for the next batch of n images; consumes `random` exactly as the reference's query() does"""
        ret, put = [], []
        for _ in range(n):
            if self.num_imgs < self.pool_size:
                ret.append(-1)
                put.append(self.num_imgs)
                self.num_imgs += 1
            elif random.uniform(0, 1) > 0.5:
                k = random.randint(0, self.pool_size - 1)
                ret.append(k)
                put.append(k)
            else:
                ret.append(-1)
                put.append(-1)
        return ret, put

    def next_batch(self, n, device):
        """draw the plan of the next batch and put it where `apply` (possibly inside a captured graph) reads it.  The upload is asynchronous
        and the host may be several steps ahead of the device: the plan travels through a small ring of pinned buffers, each guarded by
        the event of the copy that last read it."""
        if self.pool_size == 0:
            return
        ret, put = self.plan(n)
        if self._slots is None or self._slots.shape[1] != n:
            self._slots = torch.empty(2, n, dtype=torch.int32, device=device)
            self._host = [torch.empty(2, n, dtype=torch.int32).pin_memory() for _ in range(4)]
            self._sent = [None] * len(self._host)
            self._turn = 0
        k = self._turn
        self._turn = (k + 1) % len(self._host)
        if self._sent[k] is not None:
            self._sent[k].synchronize()
        self._host[k].copy_(torch.tensor([ret, put], dtype=torch.int32))
        self._slots.copy_(self._host[k], non_blocking=True)
        self._sent[k] = torch.cuda.Event()
        self._sent[k].record()

    def apply(self, name, images, out=None):
        """out = what the discriminator sees in place of `images` under the current plan; the pool store `name` is updated"""
        if self.pool_size == 0:
            return images
        st = self._stores.get(name)
        if st is not None and tuple(st.shape[1:]) != tuple(images.shape[1:]):
            # the reference would fail here (torch.cat of history entries of another shape, util/image_pool.py:56); a silently re-created
            # zero store would hand all-zero "history" images to the discriminator for every slot the plan already considers filled
            raise ValueError("ImagePool: store '%s' holds images of shape %s, the batch has %s -- a history pool cannot change its image "
                             "shape (create a new ImagePool)" % (name, tuple(st.shape[1:]), tuple(images.shape[1:])))
        if st is None:
            st = self._stores[name] = torch.zeros((self.pool_size,) + tuple(images.shape[1:]), dtype=torch.float32, device=images.device)
        if out is None:
            out = torch.empty_like(images)
        return ops.pool_query(images.contiguous(), st, self._slots[0], self._slots[1], out)

    def query(self, images):
        """the reference's one-call form (eager)"""
        if self.pool_size == 0:
            return images
        self.next_batch(images.shape[0], images.device)
        return self.apply("images", images)

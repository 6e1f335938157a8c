"""Synthetic ADE20K-shaped batches: img ~ N(0,1) [B,3,H,W] fp32 in
already-normalised space, gt in {0..C-1} with 5 % of the pixels set to the ignore index 255,
seeded per rank.  Shapes and dtypes are what the reference's collate produces
(img float32 [B,3,H,W]; gt_semantic_seg int64 [B,1,H,W], reference formating.py:206-211)."""
from __future__ import annotations

import torch


class SyntheticADE:
    def __init__(self, batch_size, size=(512, 512), num_classes=150, ignore_index=255, ignore_frac=0.05, seed=0, rank=0,
                 device='cpu', pool=4):
        self.batch_size, self.size, self.num_classes = batch_size, tuple(size), num_classes
        self.device = torch.device(device)
        g = torch.Generator().manual_seed(seed + rank)
        self._pool = []
        for _ in range(pool):  # a small pool of distinct batches, resident on the device before timing starts
            img = torch.randn(batch_size, 3, *self.size, generator=g)
            gt = torch.randint(0, num_classes, (batch_size, 1, *self.size), generator=g)
            mask = torch.rand(batch_size, 1, *self.size, generator=g) < ignore_frac
            gt[mask] = ignore_index
            self._pool.append((img.to(self.device), gt.to(self.device)))
        self._i = 0

    def next(self):
        img, gt = self._pool[self._i % len(self._pool)]
        self._i += 1
        return dict(img=img, img_metas=None, gt_semantic_seg=gt)

    __next__ = next

    def __iter__(self):
        return self

"""Adapter from a torch DataLoader to the `next()` protocol the trainer loops use (engine/data.py::SyntheticADE)."""
from __future__ import annotations

import torch


class LoaderFeed:
    """Endless batches on `device`: re-iterates the loader when an epoch ends (advancing the DistributedSampler's epoch so
    every epoch is reshuffled, as mmcv's IterBasedRunner / IterLoader does), pinned-memory async copies."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.epoch = 0
        self._it = iter(loader)

    def next(self):
        try:
            batch = next(self._it)
        except StopIteration:
            self.epoch += 1
            sampler = getattr(self.loader, 'sampler', None)
            if hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(self.epoch)
            self._it = iter(self.loader)
            batch = next(self._it)
        return dict(img=batch['img'].to(self.device, non_blocking=True), img_metas=batch.get('img_metas'),
                    gt_semantic_seg=batch['gt_semantic_seg'].to(self.device, non_blocking=True))

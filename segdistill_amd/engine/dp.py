"""Data parallelism for the KD step: one process per GPU, RCCL over xGMI.

The reference wraps the model in MMDistributedDataParallel with find_unused_parameters=True
(reference mmseg/apis/train.py:75-83): a graph walk per step plus 25 MB buckets.  The student
here is small (Segformer-B0: 15.1 MB of fp32 gradients), so the MI355X-first design is a FLAT
gradient buffer: every trainable parameter's ``.grad`` is a view into one contiguous tensor,
autograd accumulates straight into it, and ONE all-reduce (pre-scaled by 1/world) over the
fully connected xGMI fabric follows the backward (~0.03-0.2 ms, against a step of tens of ms).
No per-parameter hooks, no unused-parameter detection (the only unused parameter, SegFormerHead's
dead ``conv_seg`` -- SURVEY Q12 -- is frozen by the trainer), one memset to zero the gradients.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class DataParallelReducer:
    def __init__(self, params, world=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def zero_grad(self):
        self.flat.zero_()
        for p in self.params:  # a backward may have replaced a view (e.g. first accumulation on a None grad)
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or p.grad.data_ptr() >= self.flat.data_ptr() + self.nbytes:
                self._rebind()
                break

    def _rebind(self):
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def broadcast_parameters(self, module):
        if self.world > 1:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0)

    def all_reduce(self):
        if self.world > 1:
            self.flat.div_(self.world)
            dist.all_reduce(self.flat)

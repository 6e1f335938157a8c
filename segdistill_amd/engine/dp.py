"""Data parallelism for the KD step: one process per GPU, RCCL over xGMI.

The reference wraps the model in MMDistributedDataParallel with find_unused_parameters=True
(reference mmseg/apis/train.py:75-83): a graph walk per step plus 25 MB buckets.  The student
here is small (Segformer-B0: 15.1 MB of fp32 gradients), so the MI355X-first design is a FLAT
gradient buffer: after the backward all gradients are packed into one contiguous tensor (one
multi-tensor copy) and ONE all-reduce (pre-scaled by 1/world) over the fully connected xGMI
fabric follows (~0.03-0.2 ms, against a step of tens of ms).  No per-parameter hooks, no
unused-parameter detection (the only unused parameter, SegFormerHead's dead ``conv_seg`` --
SURVEY Q12 -- is frozen by the trainer).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def force_collectives():
    return os.environ.get('SEGDISTILL_FORCE_COLLECTIVES') == '1'


def init_distributed(backend=None):
    """Initialise from torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('SEGDISTILL_FORCE_DEVICE') is not None:
        local = int(os.environ['SEGDISTILL_FORCE_DEVICE'])
    # SEGDISTILL_FORCE_COLLECTIVES=1: create the process group and issue the gradient all-reduce even with ONE rank, so
    # that RCCL initialisation, the flat all-reduce and hipGraph capture next to RCCL's watchdog thread can be exercised on a
    # single-GPU box (tests/test_rccl_single_rank_gpu.py)
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            # SEGDISTILL_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses duplicate devices): used to exercise the
            # multi-rank GPU code path -- SyncBN, flat all-reduce, hybrid graphs -- on a single-GPU box; never for measurements
            backend = os.environ.get('SEGDISTILL_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class DataParallelReducer:
    """Gradient exchange through ONE flat buffer.

    Backward runs with ``p.grad = None`` so autograd simply hands each parameter its gradient tensor (pre-set ``.grad``
    views would make it launch one tiny in-place add per parameter: ~190 extra kernels per step for Segformer-B0).
    With world > 1 the gradients are then packed into the flat buffer by one multi-tensor copy, averaged by ONE
    all-reduce (pre-scaled by 1/world) and ``.grad`` is re-pointed at views of the buffer; with world == 1 nothing
    at all happens after the backward."""

    def __init__(self, params, world=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.collective = self.world > 1 or (force_collectives() and dist.is_initialized())
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self.views = []
        self._strided = []          # parameters kept in a non-default dense layout (channels-last conv filters)
        off = 0
        for p in self.params:
            n = p.numel()
            self.views.append(self._view_like(self.flat[off:off + n], p))
            if not p.is_contiguous():
                self._strided.append(p)
            off += n

    @staticmethod
    def _view_like(segment, p):
        """A view of the flat segment with p's shape AND strides: the fused optimizer kernels require parameter, gradient and
        state tensors of identical layout."""
        if p.is_contiguous():
            return segment.view_as(p)
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last):
            n, c, h, w = p.shape
            return segment.view(n, h, w, c).permute(0, 3, 1, 2)
        raise ValueError(f'parameter of shape {tuple(p.shape)} with strides {p.stride()} is neither contiguous nor channels-last')

    def match_layouts(self):
        """Gradients that autograd produced in another dense layout than their parameter (possible for channels-last filters)
        are re-laid-out; a no-op -- a few stride comparisons -- in the common case."""
        for p in self._strided:
            if p.grad is not None and p.grad.stride() != p.stride():
                p.grad = p.grad.contiguous(memory_format=torch.channels_last)

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def broadcast_parameters(self, module):
        if self.collective:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0)

    def pack(self):
        """Gradients -> the flat buffer (one multi-tensor copy), pre-scaled by 1/world.  Device work only and no host decision
        that depends on values: a trainer that captures the step records it at the end of the backward graph, reading
        the capture's static gradient tensors."""
        grads, views = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad is not v:
                grads.append(p.grad)
                views.append(v)
        if grads:
            torch._foreach_copy_(views, grads)
        self.flat.div_(self.world)

    def exchange(self):
        """ONE all-reduce of the packed buffer; `.grad` then points at its views."""
        dist.all_reduce(self.flat)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self):
        if not self.collective:
            self.match_layouts()
            return
        self.pack()
        self.exchange()

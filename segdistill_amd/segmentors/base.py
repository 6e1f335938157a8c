"""Segmentor base: ``forward`` dispatch, ``train_step`` runner contract and loss parsing.

Counterpart of reference mmseg/models/segmentors/base.py (forward :113-126, train_step
:128-162, _parse_losses :174-209).  The runner contract is unchanged:
``train_step(data_batch, optimizer) -> {'loss': tensor, 'log_vars': {str: float}, 'num_samples': int}``.

MI355X difference: the reference all-reduces and ``.item()``s every log variable separately
(4-6 NCCL calls + 4-6 host syncs per step, :204-207).  Here all log scalars are stacked into
ONE tensor, reduced with ONE RCCL all-reduce and copied to the host with ONE sync; with
``defer_log_sync`` both the all-reduce and the host copy happen only on logging iterations.
"""
from __future__ import annotations

from abc import ABCMeta, abstractmethod
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn


def _mean(value):
    """value.mean() of a large GPU map as two single-block-per-output sums.  ATen's one-pass mean of >= 2^15 elements is a multi-block reduction
    whose semaphores are zeroed by hipMemsetAsync -- in a captured step the ONLY memset node left (the profiler finds `aten::mean` and nothing
    else), and a memset node of a replayed hipGraph is not safe on ROCm 7.x: an eager hipMemsetAsync issued between two replays (a `torch.equal`,
    an evaluation hook) can leave the node zeroing nothing from then on -- the logged `decode.loss_seg` then read 0.0115 instead of 4.82 on every
    later step (tests/test_graph_gpu.py).  Plain differentiable ops: the gradient arrives as ONE value broadcast over the map
    (all strides 0), which the fused CE backward takes as a scalar instead of reading a dense 1/N map."""
    if value.is_cuda and value.numel() >= 8192 and value.is_contiguous() and value.is_floating_point():
        for group in (1024, 512, 256):
            if value.numel() % group == 0:
                # (sums, then ONE division: a mean's backward divides the EXPANDED gradient, i.e. materialises the dense map again)
                return value.reshape(-1, group).sum(1).sum() / value.numel()
    return value.mean()


def parse_losses(losses, want_host_values=True, extra=None):
    """-> (loss tensor, OrderedDict name -> float | 0-dim tensor).

    mean of every entry; ``loss`` = sum of the entries whose key contains 'loss'
    (so acc_seg is excluded and the KD keys are included).  With ``want_host_values=False`` (``defer_log_sync``) the
    returned 0-dim tensors are this RANK's values: nobody reads them until a log line is due, so their cross-rank mean
    is taken there (``KDTrainer.log_values`` -- one collective per log interval instead of one per step, and none
    inside a captured step).  ``extra(log_vars) -> {name: 0-dim tensor}`` adds entries after the sum is formed (reference
    SD_structure.py:124-134 puts `deg` there, in front of `loss`)."""
    log_vars = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = _mean(value)
        elif isinstance(value, list):
            log_vars[name] = sum(_mean(v) for v in value)
        else:
            raise TypeError(f'{name} is not a tensor or list of tensors')
    terms = [v for k, v in log_vars.items() if 'loss' in k]
    loss = terms[0] if terms else 0
    for v in terms[1:]:      # (Python's sum() starts from the int 0: one more add kernel)
        loss = loss + v
    if extra is not None:   # diagnostics computed from the per-key means (SDModule's `log_grad` angle): logged, never part of `loss`
        log_vars.update(extra(log_vars))
    log_vars['loss'] = loss
    names = list(log_vars)
    packed = torch.stack([log_vars[n].detach().float().reshape(()) for n in names])
    if want_host_values and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        packed = packed / dist.get_world_size()
        dist.all_reduce(packed)
    if want_host_values:
        host = packed.tolist()  # the single device->host sync of the step
        out = OrderedDict(zip(names, host))
    else:
        out = OrderedDict((n, packed[i]) for i, n in enumerate(names))
    return loss, out


class BaseSegmentor(nn.Module, metaclass=ABCMeta):
    def __init__(self):
        super().__init__()
        self.fp16_enabled = False
        self.defer_log_sync = False  # True: log_vars hold 0-dim device tensors, no host sync in train_step

    with_neck = property(lambda self: getattr(self, 'neck', None) is not None)
    with_auxiliary_head = property(lambda self: getattr(self, 'auxiliary_head', None) is not None)
    with_decode_head = property(lambda self: getattr(self, 'decode_head', None) is not None)

    @abstractmethod
    def forward_train(self, img, img_metas, **kwargs):
        ...

    def init_weights(self, pretrained=None):
        pass

    def forward_test(self, imgs, img_metas, **kwargs):
        raise NotImplementedError('test-time inference (slide / aug) is outside the KD train-step path')

    def forward(self, img, img_metas=None, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def _parse_losses(self, losses):
        return parse_losses(losses, want_host_values=not self.defer_log_sync)

    def train_step(self, data_batch, optimizer=None, **kwargs):
        losses = self(**data_batch)
        loss, log_vars = self._parse_losses(losses)
        img = data_batch['img']
        return dict(loss=loss, log_vars=log_vars, num_samples=len(getattr(img, 'data', img)))

    def val_step(self, data_batch, **kwargs):
        return self(**data_batch, **kwargs)

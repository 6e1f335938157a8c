"""Supervised segmentation loss of the student: softmax cross-entropy with ignore_index
and the top-k accuracy log variable.

Counterpart of reference mmseg/models/losses/cross_entropy_loss.py (cross_entropy :9-32,
CrossEntropyLoss :139-197), losses/utils.py (weight_reduce_loss :26-60) and
losses/accuracy.py (accuracy :4-49).  Only the softmax form is on the KD path; the
sigmoid / mask variants raise.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..builder import LOSSES


def _reduce(loss, weight, reduction, avg_factor):
    if weight is not None:
        if weight.dim() != loss.dim():
            raise AssertionError('weight and loss must have the same number of dims')
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        if reduction == 'none':
            return loss
        raise ValueError(reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def cross_entropy(pred, label, weight=None, class_weight=None, reduction='mean', avg_factor=None, ignore_index=-100):
    """Per-pixel CE (ignored pixels contribute 0 and still count in a plain mean, as in
    the reference) followed by the weight / reduction step."""
    per_pixel = F.cross_entropy(pred, label, weight=class_weight, reduction='none', ignore_index=ignore_index)
    return _reduce(per_pixel, None if weight is None else weight.float(), reduction, avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0):
        super().__init__()
        if use_sigmoid or use_mask:
            raise NotImplementedError('sigmoid / mask cross-entropy are not on the KD train-step path')
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction = reduction
        self.class_weight = class_weight
        self.loss_weight = loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if reduction_override not in (None, 'none', 'mean', 'sum'):
            raise AssertionError(reduction_override)
        cw = None if self.class_weight is None else cls_score.new_tensor(self.class_weight)
        return self.loss_weight * cross_entropy(cls_score, label, weight, class_weight=cw,
                                                reduction=reduction_override or self.reduction, avg_factor=avg_factor, **kwargs)


def accuracy(pred, target, topk=1, thresh=None):
    """Top-k accuracy in percent over ALL target elements (ignored pixels count as misses,
    reference accuracy.py:47 divides by target.numel())."""
    single = isinstance(topk, int)
    ks = (topk,) if single else tuple(topk)
    if pred.size(0) == 0:
        out = [pred.new_tensor(0.) for _ in ks]
        return out[0] if single else out
    assert pred.ndim == target.ndim + 1 and pred.size(0) == target.size(0)
    kmax = max(ks)
    assert kmax <= pred.size(1), f'maxk {kmax} exceeds pred dimension {pred.size(1)}'
    if kmax == 1 and thresh is None:
        hit = (pred.argmax(dim=1) == target).unsqueeze(0)
    else:
        val, idx = pred.topk(kmax, dim=1)
        hit = idx.transpose(0, 1).eq(target.unsqueeze(0).expand(kmax, *target.shape))
        if thresh is not None:
            hit = hit & (val.transpose(0, 1) > thresh)
    out = [hit[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.numel()) for k in ks]
    return out[0] if single else out

"""torch-eager stand-ins for the product's HIP criteria, used ONLY by bench.py's
``cpu_baseline`` leg and by tests (TEST INFRASTRUCTURE -- see oracle/__init__.py).

``swap_in_eager_criteria(model)`` replaces every criterion inside a built SDModule's
DistillationLoss by an op-for-op eager restatement of the reference's KLDLoss.forward
(reference mmseg/models/distillation/losses.py:95-113) so that the whole KD train step can
run on host cores -- this is the "port" CPU baseline reported next to the MI355X number.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import kd_ref


class EagerKLD(nn.Module):
    def __init__(self, alpha, tau, resize_config, shuffle_config, transform_config, warmup_config, earlydecay_config):
        super().__init__()
        self.schedule = kd_ref.AlphaSchedule(alpha, warmup_config, earlydecay_config)
        self.tau = tau
        self.resize_config, self.shuffle_config, self.transform_config = resize_config, shuffle_config, transform_config

    @property
    def alpha(self):
        return self.schedule.alpha

    def forward(self, x_student, x_teacher, gt, n_iter):
        alpha = self.schedule.step(n_iter)
        out_size = None
        if self.resize_config:
            ref = x_teacher if self.resize_config.get('target', 'gt') == 'teacher' else gt
            out_size = tuple(ref.shape[2:])
        perm = None
        if self.shuffle_config and n_iter % self.shuffle_config['interval'] == 0:
            perm = torch.randperm(x_student.shape[1])
        kind = self.transform_config['loss_type'] if self.transform_config else None
        g = self.transform_config.get('group_size') if self.transform_config else None
        return kd_ref.eager_kld(x_student, x_teacher, alpha=alpha, tau=self.tau, out_size=out_size, perm=perm, loss_type=kind,
                                group_size=g)


class EagerAT(nn.Module):
    def forward(self, s, t, gt, step):
        return kd_ref.eager_at(s, t)


class EagerIFVD(nn.Module):
    def forward(self, s, t, gt, step):
        return kd_ref.eager_ifvd(s, t, gt)


def eager_twin(criterion):
    name = type(criterion).__name__
    if name == 'ATLoss':
        return EagerAT()
    if name == 'IFVDLoss':
        return EagerIFVD()
    return EagerKLD(criterion.alpha_0, criterion.tau, criterion.resize_config, criterion.shuffle_config, criterion.transform_config,
                    criterion.warmup_config, criterion.earlydecay_config)


def swap_in_eager_criteria(model):
    dl = model.distillation_loss
    for i, c in enumerate(list(dl.criteria)):
        dl.criteria[i] = eager_twin(c)
    for key, align in dl.aligns.items():
        w, b = align.weight, align.bias
        conv = nn.Conv2d(w.shape[1], w.shape[0], 1, bias=b is not None)
        with torch.no_grad():
            conv.weight.copy_(w.reshape(w.shape[0], w.shape[1], 1, 1))
            if b is not None:
                conv.bias.copy_(b)
        dl.aligns[key] = conv
    return model

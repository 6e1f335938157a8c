"""Feature taps and loss dispatch of the KD plug-in layer.

Interface counterpart of reference mmseg/models/distillation/opts.py: Extractor :13-71
(forward hooks by dotted module name, stored only while training :66-71) and
DistillationLoss :74-112 (criterion construction :78-84, call :100-103, key naming :105-110).

Differences, all deliberate:
 * Q7  the criterion class is looked up in the DISTILL_LOSSES registry instead of ``eval(loss_name)``;
 * Q3  two entries on the same layer pair no longer overwrite each other: a colliding key gets a
       ``#k`` suffix;
 * Q10 token-major taps ``[B, N, C]`` (e.g. ``decode_head.linear_c1``) are viewed as ``[B, C, h, w]``
       instead of crashing;
 * a-15 ``channel_nums=(Cs, Ct)`` (documented at reference opts.py:25-27 but absent from its live
       code) inserts a trainable 1x1 projection of the student feature, run by the MFMA kernel.
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from .. import ops
from ..builder import DISTILL_LOSSES


def _as_list(v):
    return list(v) if isinstance(v, (list, tuple)) else [v]


class Extractor(nn.Module):
    def __init__(self, student, teacher, distillation, verbose=False):
        super().__init__()
        self.student_features = {}
        self.teacher_features = {}
        want_s, want_t = [], []
        for entry in distillation:
            want_s += _as_list(entry['student_layer'])
            want_t += _as_list(entry['teacher_layer'])
        self.hooked = {'student': [], 'teacher': []}
        for kind, net, wanted in (('teacher', teacher, want_t), ('student', student, want_s)):
            for name, module in net.named_modules():
                if name in wanted:
                    module.register_forward_hook(partial(self._store, name=name, kind=kind))
                    self.hooked[kind].append(name)
                    if verbose:
                        print(f'{kind}_layer :{name} hooked!!!!')
            missing = sorted(set(wanted) - set(self.hooked[kind]))
            if missing:
                raise KeyError(f'{kind} has no module named {missing}; taps must be dotted module names')

    def _store(self, module, inputs, output, name, kind):
        if self.training:
            (self.student_features if kind == 'student' else self.teacher_features)[name] = output

    def clear(self):
        self.student_features.clear()
        self.teacher_features.clear()


class FeatureAlign(nn.Module):
    """Trainable 1x1 projection Cs -> Ct of the STUDENT feature (SURVEY a-15).  Lives with the
    distillation loss so it is in the optimizer and in the gradient all-reduce."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_normal_(self.weight.view(out_channels, in_channels, 1, 1), mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        return ops.align1x1(x, self.weight, self.bias)

    def extra_repr(self):
        return f'{self.weight.shape[1]} -> {self.weight.shape[0]}'


def _to_nchw(x):
    """[B,C,h,w] stays; token-major [B,N,C] becomes [B,C,h,w] with h=w=sqrt(N)."""
    if x.dim() == 4:
        return x
    if x.dim() == 3:
        b, n, c = x.shape
        side = math.isqrt(n)
        if side * side != n:
            raise ValueError(f'cannot view {n} tokens as a square map')
        return x.transpose(1, 2).reshape(b, c, side, side)
    raise ValueError(f'tapped feature must be 3-D or 4-D, got {tuple(x.shape)}')


class DistillationLoss(nn.Module):
    def __init__(self, distillation):
        super().__init__()
        self.criteria = nn.ModuleList()
        self.aligns = nn.ModuleDict()
        for i, entry in enumerate(distillation):
            name = entry['loss_name']
            cfg = entry['loss_config']
            if isinstance(cfg, tuple):  # a trailing comma in a config file makes it a 1-tuple (reference :81-82)
                cfg = cfg[0]
            cls = DISTILL_LOSSES.get(name)
            if cls is None:
                raise KeyError(f'{name} is not a registered distillation loss; known: {sorted(DISTILL_LOSSES.module_dict)}')
            criterion = cls(**cfg)
            entry['criterion'] = criterion
            self.criteria.append(criterion)
            if entry.get('channel_nums'):
                cs, ct = entry['channel_nums']
                self.aligns[str(i)] = FeatureAlign(cs, ct)
        self.distillation = distillation

    def set_graph_safe(self, flag=True):
        for c in self.criteria:
            if hasattr(c, 'graph_safe'):
                c.graph_safe = flag

    def prepare_replay(self, step):
        for c in self.criteria:
            if hasattr(c, 'prepare_replay'):
                c.prepare_replay(step)

    def forward(self, student_features, teacher_features, gt_semantic_seg, step, student=None, teacher=None):
        out = {}
        for i, entry in enumerate(self.distillation):
            s_name, t_name = entry['student_layer'], entry['teacher_layer']
            if isinstance(s_name, list):
                raise NotImplementedError('list-typed layers (attention-pair criteria) are not used by any shipped config')
            x_s, x_t = _to_nchw(student_features[s_name]), _to_nchw(teacher_features[t_name])
            if str(i) in self.aligns:
                x_s = self.aligns[str(i)](x_s)
            loss = self.criteria[i](x_s, x_t, gt_semantic_seg, step)
            try:
                info = entry['loss_config']['transform_config']
            except (KeyError, TypeError):
                info = 'other'
            key = f'loss_{s_name}<->{t_name}_{info}'
            if key in out:
                k = 1
                while f'{key}#{k}' in out:
                    k += 1
                key = f'{key}#{k}'
            out[key] = loss
        return out

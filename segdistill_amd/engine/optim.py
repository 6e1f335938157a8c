"""Optimizer construction and LR schedule with the semantics the reference's configs rely on
(mmcv DefaultOptimizerConstructor + PolyLrUpdaterHook; used by e.g.
reference local_configs/exp_tab5/segformer_CGD.py:60-70):

* ``paramwise_cfg.custom_keys``: keys sorted alphabetically then by length (desc); the first key
  that is a SUBSTRING of a parameter's dotted name sets lr = base_lr*lr_mult and
  weight_decay = base_wd*decay_mult for that parameter;
* poly schedule ``lr = (base - min_lr) * (1 - it/max_iters)**power + min_lr`` with linear warm-up
  ``lr * (1 - (1 - it/warmup_iters) * (1 - warmup_ratio))`` for it < warmup_iters.

Frozen parameters (the teacher) are simply left out of the optimizer.
"""
from __future__ import annotations

import torch


def build_optimizer(model, cfg):
    cfg = dict(cfg)
    kind = cfg.pop('type')
    paramwise = cfg.pop('paramwise_cfg', None) or {}
    custom = paramwise.get('custom_keys', {})
    keys = sorted(sorted(custom.keys()), key=len, reverse=True)
    base_lr, base_wd = cfg['lr'], cfg.get('weight_decay', 0.0)
    buckets = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        lr, wd = base_lr, base_wd
        for k in keys:
            if k in name:
                lr = base_lr * custom[k].get('lr_mult', 1.0)
                wd = base_wd * custom[k].get('decay_mult', 1.0)
                break
        buckets.setdefault((lr, wd), []).append(p)
    groups = [dict(params=ps, lr=lr, weight_decay=wd, initial_lr=lr) for (lr, wd), ps in buckets.items()]
    cls = getattr(torch.optim, kind)
    extra = {}
    if kind in ('AdamW', 'Adam', 'SGD') and all(p.is_cuda for g in groups for p in g['params']):
        extra['fused'] = True  # one multi-tensor kernel per group instead of ~10 tiny kernels per parameter
    cfg.pop('lr', None)
    cfg.pop('weight_decay', None)
    return cls(groups, lr=base_lr, weight_decay=base_wd, **cfg, **extra)


class PolyLR:
    def __init__(self, optimizer, max_iters, power=1.0, min_lr=0.0, warmup=None, warmup_iters=0, warmup_ratio=0.1, **_ignored):
        self.opt, self.max_iters, self.power, self.min_lr = optimizer, max_iters, power, min_lr
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])

    def lr_at(self, base, it):
        coeff = (1 - it / self.max_iters) ** self.power
        lr = (base - self.min_lr) * coeff + self.min_lr
        if self.warmup == 'linear' and it < self.warmup_iters:
            lr *= 1 - (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
        elif self.warmup == 'constant' and it < self.warmup_iters:
            lr *= self.warmup_ratio
        elif self.warmup == 'exp' and it < self.warmup_iters:
            lr *= self.warmup_ratio ** (1 - it / self.warmup_iters)
        return lr

    def step(self, it):
        for g in self.opt.param_groups:
            g['lr'] = self.lr_at(g['initial_lr'], it)

"""One KD training iteration, the way the reference's runner drives it (mmcv IterBasedRunner +
OptimizerHook: ``zero_grad -> train_step -> loss.backward -> step``, reference
mmseg/apis/train.py:97-138 and SURVEY.md Appendix B), on the MI355X-first DP substrate."""
from __future__ import annotations

import torch

from .dp import DataParallelReducer
from .optim import PolyLR, build_optimizer


def _freeze_dead_parameters(model):
    """SegFormerHead never uses the conv_seg it inherits: keep it in the state dict,
    keep it out of the optimizer / reducer."""
    from ..decode_heads import SegFormerHead
    for m in model.modules():
        if isinstance(m, SegFormerHead):
            for p in m.conv_seg.parameters():
                p.requires_grad = False


class KDTrainer:
    def __init__(self, model, optimizer_cfg, lr_cfg=None, max_iters=160000, world=1, log_interval=50, precision=None):
        self.model = model
        _freeze_dead_parameters(model)
        self.optimizer = build_optimizer(model, optimizer_cfg)
        self.reducer = DataParallelReducer([p for g in self.optimizer.param_groups for p in g['params']], world=world)
        self.reducer.broadcast_parameters(model)
        lr_cfg = dict(lr_cfg or {})
        lr_cfg.pop('policy', None)
        lr_cfg.pop('by_epoch', None)
        self.sched = PolyLR(self.optimizer, max_iters=max_iters, **lr_cfg) if lr_cfg is not None else None
        self.iter = 0
        self.log_interval = log_interval
        self.last_log_vars = None
        model.defer_log_sync = True  # host sync only when a log line is due
        # precision=dict(activations='bf16'): bf16 storage of activations / tapped features (autocast), fp32 master weights,
        # fp32 accumulation inside every HIP kernel (BASELINE config 5)
        self.bf16 = bool(precision) and precision.get('activations') == 'bf16'

    def step(self, batch):
        self.model.train()
        if self.sched is not None:
            self.sched.step(self.iter)
        self.reducer.zero_grad()
        if self.bf16 and batch['img'].is_cuda:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out = self.model.train_step(batch, self.optimizer)
        else:
            out = self.model.train_step(batch, self.optimizer)
        out['loss'].backward()
        self.reducer.all_reduce()
        self.optimizer.step()
        self.iter += 1
        self.last_log_vars = out['log_vars']
        return out

    def log_values(self):
        """Host copies of the most recent log variables (one device->host sync)."""
        if self.last_log_vars is None:
            return {}
        names = list(self.last_log_vars)
        vals = torch.stack([torch.as_tensor(self.last_log_vars[n]).float().reshape(()) for n in names]).tolist()
        return dict(zip(names, vals))

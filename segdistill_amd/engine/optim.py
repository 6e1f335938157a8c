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

import os

import numpy as np
import torch


class HipAdamW(torch.optim.AdamW):
    """torch.optim.AdamW (same constructor, param_groups, state and state_dict) whose step() is ONE launch of sd_adamw_multi over every
    tensor that has a gradient (csrc/optim.hip) instead of one multi-tensor launch per (group, chunk list): 0.25 ms -> ~0.03 ms of GPU time per
    step for Segformer-B0 (689 -> 707 imgs/s on the captured step).  fp32 parameters on the GPU, no amsgrad / maximize.
    Host side: the descriptor table lives in a numpy record array; a step whose gradient TENSORS are the ones of the previous step (graph
    replay, or gradients living in the data-parallel flat buffer) reuses the device copy after ~200 identity checks; new gradient tensors
    (eager backward) cost one column update and a 10 KB upload; only a change of the set of tensors rebuilds it.
    NOTE ``optimizer.state[p]['step']`` is NOT advanced by step(): the kernel keeps the per-tensor step counts on the device and the host
    tensors are brought up to date only by ``state_dict()`` (checkpointing goes through it).  Code that reads ``state[p]['step']`` directly
    between checkpoints sees the count of the last ``state_dict()`` / ``load_state_dict()`` call."""
    _DESC = np.dtype([('p', '<u8'), ('g', '<u8'), ('m', '<u8'), ('v', '<u8'), ('s', '<u8'), ('wd', '<f4'), ('gm', '<i4'), ('n', '<i8')])

    def __init__(self, params, **kw):
        kw.pop('fused', None)
        kw.pop('foreach', None)
        super().__init__(params, foreach=False, fused=False, **kw)
        assert self._DESC.itemsize == 56
        self._global = 0            # optimizer steps taken
        self._missed = None         # id(parameter) -> steps it took no part in (torch counts steps per tensor); None: derive from the state
        self._live_params = None    # parameters of the current table, in table order
        self._live_grads = None     # their gradient tensors at the last upload
        self._absent = []           # tracked tensors outside the current table
        self._desc = None
        self._dev_tensors = self._dev_blocks = None
        self._nblocks = 0
        self._planes = None         # planes.Refresher, created at the first step

    @staticmethod
    def _shadow_of(p):
        """The bf16 shadow a low-precision forward keeps on the parameter (segdistill_amd.linear.lowp_copy): valid only for the parameter
        version it was made from and for the parameter's own dense layout."""
        sh = getattr(p, '_sd_shadow', None)
        if sh is None or sh[0] != p._version or sh[1].stride() != p.stride() or sh[1].dtype != torch.bfloat16:
            return None
        return sh[1]

    def _derive_counts(self):
        steps = {id(p): int(float(st['step'])) for p, st in self.state.items() if 'step' in st}
        self._global = max(steps.values(), default=0)
        self._missed = {k: self._global - v for k, v in steps.items()}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._missed = None
        self._live_params = None

    def state_dict(self):
        if self._missed is not None:
            for p, st in self.state.items():
                if 'step' in st:
                    st['step'] = torch.tensor(float(self._global - self._missed.get(id(p), self._global)), dtype=torch.float32)
        return super().state_dict()

    def _rebuild(self, live, L, dev):
        chunk = L.sd_adamw_chunk()
        desc = np.zeros(len(live), dtype=self._DESC)
        blocks = []
        for ti, (gi, p) in enumerate(live):
            st = self.state[p]
            if not st:
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                self._missed[id(p)] = self._global - 1          # first seen in this step (already counted in _global)
            if st['exp_avg'].stride() != p.stride() or st['exp_avg_sq'].stride() != p.stride():
                raise RuntimeError('HipAdamW: optimizer state laid out differently from its parameter')
            missed = self._missed[id(p)]
            if missed >= 1 << 23:
                raise RuntimeError('HipAdamW: a tensor skipped too many steps')
            sh = self._shadow_of(p)
            desc[ti] = (p.data_ptr(), 0, st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), 0 if sh is None else sh.data_ptr(),
                        float(self.param_groups[gi]['weight_decay']), gi | (missed << 8), p.numel())
            blocks += [(ti, c) for c in range((p.numel() + chunk - 1) // chunk)]
        self._desc = desc
        self._dev_blocks = torch.tensor(blocks, dtype=torch.int32).view(-1).to(dev)
        self._nblocks = len(blocks)
        self._live_params = [p for _, p in live]
        self._live_shadows = [self._shadow_of(p) for _, p in live]
        self._live_key = tuple((gi, float(self.param_groups[gi]['weight_decay'])) for gi, _ in live)
        self._live_grads = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        import ctypes

        from .. import _lib
        from ..ops import _stream_ptr
        live = [(gi, p) for gi, g in enumerate(self.param_groups) for p in g['params'] if p.grad is not None]
        if not live:
            return loss
        L = _lib.lib()
        dev = live[0][1].device
        if len(self.param_groups) > L.sd_adamw_max_groups():
            raise RuntimeError(f'HipAdamW: at most {L.sd_adamw_max_groups()} parameter groups')
        b1, b2 = self.param_groups[0]['betas']
        eps = self.param_groups[0]['eps']
        for g in self.param_groups[1:]:
            if tuple(g['betas']) != (b1, b2) or g['eps'] != eps:
                raise RuntimeError('HipAdamW: betas / eps must be the same in every parameter group')
        if self._missed is None:
            self._derive_counts()
        self._global += 1
        same_set = (self._live_params is not None and len(live) == len(self._live_params) and all(p is q for (_, p), q in zip(live, self._live_params))
                    and all(self._shadow_of(p) is s for (_, p), s in zip(live, self._live_shadows))
                    and self._live_key == tuple((gi, float(self.param_groups[gi]['weight_decay'])) for gi, _ in live))
        if same_set:
            for k in self._absent:      # tensors that sit this step out fall one step behind (torch counts steps per tensor)
                self._missed[k] += 1
        else:
            present = {id(p) for _, p in live}
            self._absent = [k for k in self._missed if k not in present]
            for k in self._absent:
                self._missed[k] += 1
            self._rebuild(live, L, dev)
        if self._live_grads is None or any(p.grad is not g for p, g in zip(self._live_params, self._live_grads)):
            for p in self._live_params:
                gr = p.grad
                if gr.dtype != torch.float32 or gr.is_sparse:
                    raise RuntimeError('HipAdamW: fp32 dense gradients only')
                if gr.stride() != p.stride():
                    cl = p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last)
                    p.grad = gr.contiguous(memory_format=torch.channels_last) if cl else gr.contiguous()
            self._live_grads = [p.grad for p in self._live_params]
            self._desc['g'] = [g.data_ptr() for g in self._live_grads]
            # a fresh device buffer per upload: a step still in flight on the stream keeps reading the previous one
            self._dev_tensors = torch.from_numpy(self._desc.view(np.uint8).reshape(-1).copy()).to(dev, non_blocking=True)
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g['lr']) for g in self.param_groups])
        _lib.check(L.sd_adamw_multi(self._dev_tensors.data_ptr(), self._dev_blocks.data_ptr(), self._nblocks, lrs, len(self.param_groups), float(b1),
                                    float(b2), float(eps), self._global, _stream_ptr()), 'sd_adamw_multi')
        # the parameters were written through raw pointers (no version bump): rewrite, in ONE launch, the pre-split bf16 planes that the
        # split-bf16 GEMMs keep of them (segdistill_amd/planes.py; buffers stay where they are, so a captured graph keeps reading them)
        if self._planes is None:
            from ..planes import Refresher
            self._planes = Refresher()
        self._planes.refresh(self._live_params)
        return loss


def sync_derived_weights(params):
    """Copies of the weights that kernels read instead of the fp32 parameter -- pre-split bf16 planes (planes.py), bf16 shadows
    (linear.lowp_copy) -- rewritten IN PLACE from the parameters' current values: one launch for all planes, one multi-tensor copy for all shadows."""
    from .. import planes
    params = [p for p in params if p.is_cuda]
    if not params:
        return
    planes.sync(params)
    with torch.no_grad():
        # the bf16 shadows in ONE multi-tensor copy (a per-parameter `copy_` was one tiny launch per weight and bias: hundreds per step under
        # bf16 storage on the torch-optimizer paths)
        src, dst = [], []
        for p in params:
            sh = getattr(p, '_sd_shadow', None)
            if sh is not None:
                src.append(p.detach())
                dst.append(sh[1])
                p._sd_shadow = (p._version, sh[1])
        if dst:
            torch._foreach_copy_(dst, src)


def _refresh_after_step(optimizer, *_args, **_kw):
    sync_derived_weights([p for g in optimizer.param_groups for p in g['params']])


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
    if (kind == 'AdamW' and os.environ.get('SEGDISTILL_HIP_ADAMW', '1') == '1' and not cfg.get('amsgrad', False) and not cfg.get('maximize', False)
            and all(p.is_cuda and p.dtype == torch.float32 for g in groups for p in g['params'])):
        cfg.pop('lr', None)
        cfg.pop('weight_decay', None)
        return HipAdamW(groups, lr=base_lr, weight_decay=base_wd, **cfg)
    cls = getattr(torch.optim, kind)
    extra = {}
    if kind in ('AdamW', 'Adam', 'SGD') and all(p.is_cuda for g in groups for p in g['params']):
        extra['fused'] = True  # one multi-tensor kernel per group instead of ~10 tiny kernels per parameter
    cfg.pop('lr', None)
    cfg.pop('weight_decay', None)
    opt = cls(groups, lr=base_lr, weight_decay=base_wd, **cfg, **extra)
    # Only HipAdamW rewrites the planes / shadows itself.  Any torch optimizer gets a step post-hook that does: torch's FUSED optimizers
    # (what `extra` asks for) do NOT bump the parameters' in-place version counters, so the "fresh iff versions agree" rule of planes.py
    # never fires for them -- eager steps kept multiplying by the first step's planes (found in round 4 with tests/test_graph_gpu.py; ADVICE
    # r3 had flagged the graph-replay half of it, where no Python forward runs at all).  One launch per step.
    opt.register_step_post_hook(_refresh_after_step)
    return opt


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

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


class TapsComplete(Exception):
    """Raised by the teacher's LAST tap once every tapped teacher feature of the step is in hand (Extractor.stop_teacher_after_taps): the frozen
    network's remaining layers feed nothing -- the caller (SDModule._teacher_forward) catches it and skips them."""


class Extractor(nn.Module):
    def __init__(self, student, teacher, distillation, verbose=False):
        super().__init__()
        self.student_features = {}
        self.teacher_features = {}
        self.stop_teacher_after_taps = False     # armed by SDModule around a frozen teacher's forward-only pass
        want_s, want_t = [], []
        for entry in distillation:
            want_s += _as_list(entry['student_layer'])
            want_t += _as_list(entry['teacher_layer'])
        self.hooked = {'student': [], 'teacher': []}
        self._hook_ids = {}
        for kind, net, wanted in (('teacher', teacher, want_t), ('student', student, want_s)):
            for name, module in net.named_modules():
                if name in wanted:
                    handle = module.register_forward_hook(partial(self._store, name=name, kind=kind))
                    self._hook_ids[(kind, name)] = handle.id
                    self.hooked[kind].append(name)
                    if verbose:
                        print(f'{kind}_layer :{name} hooked!!!!')
            missing = sorted(set(wanted) - set(self.hooked[kind]))
            if missing:
                raise KeyError(f'{kind} has no module named {missing}; taps must be dotted module names')

    def _store(self, module, inputs, output, name, kind):
        if self.training:
            (self.student_features if kind == 'student' else self.teacher_features)[name] = output
            if kind == 'teacher' and self.stop_teacher_after_taps and len(self.teacher_features) == len(set(self.hooked['teacher'])):
                # hooks that were registered on this module AFTER ours (somebody else watching the same tap) still see the call
                mine, after = self._hook_ids.get((kind, name)), False
                for hid, hook in list(module._forward_hooks.items()):
                    if after:
                        hook(module, inputs, output)
                    after = after or hid == mine
                raise TapsComplete

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

    def forward_tokens(self, x):
        """The same projection on a token-major tap [B, N, Cs] -> [B, N, Ct]: in this layout a 1x1 conv IS a Linear over the tokens, so it
        runs through the token Linear machinery (library / MFMA GEMM forward, split-K MFMA weight gradient, deferred bias column sums) and
        nothing is transposed.  bf16 taps: bf16 products with fp32 accumulation, fp32 master weight (as under autocast)."""
        from ..linear import token_linear
        if x.dtype == torch.bfloat16 and not torch.is_autocast_enabled():
            with torch.autocast('cuda', dtype=torch.bfloat16):
                return token_linear(x, self.weight, self.bias, defer_ok=True)
        return token_linear(x, self.weight, self.bias, defer_ok=True)

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

    def _token_form(self, i, x_student, x_teacher):
        """Entry i stays token-major ([B,N,C] taps as decode_head.linear_c1..4 emit them) when its criterion has a token form for them
        (KLDLoss 'channel' rows without a resize: csrc/cgd_tok.hip) -- no [B,C,h,w] view, hence no transpose copy of either tap or of the
        gradient."""
        crit = self.criteria[i]
        if not (x_student.dim() == 3 and x_teacher.dim() == 3 and hasattr(crit, 'tokens_ok')):
            return False
        align = self.aligns[str(i)] if str(i) in self.aligns else None
        cs_out = align.weight.shape[0] if align is not None else x_student.shape[2]
        return cs_out == x_teacher.shape[2] and x_student.shape[:2] == x_teacher.shape[:2] and crit.tokens_ok(x_teacher)

    def entry_loss(self, i, x_student, x_teacher, gt_semantic_seg, step):
        """Entry i of the config on RAW taps ([B,C,h,w], or token-major [B,N,C]): align projection of the student feature (when the entry has
        channel_nums) + criterion.  Token-major taps without a token form are viewed as [B,C,h,w] (reference opts.py:25-27)."""
        crit = self.criteria[i]
        align = self.aligns[str(i)] if str(i) in self.aligns else None
        if self._token_form(i, x_student, x_teacher):
            if align is not None and ops.align_cgd_tokens_supported(x_student, align.weight, x_teacher):
                # projection and criterion in one pass each way (csrc/align_tok.hip): the projected feature is never written
                if hasattr(crit, 'host_prepare') and not (x_student.is_cuda and torch.cuda.is_current_stream_capturing()):
                    crit.host_prepare(step, align.weight.shape[0])
                meta, alpha_t = crit.token_job(x_teacher, step)
                loss = ops.align_cgd_tokens_multi([(x_student, align.weight, align.bias, x_teacher)], [meta])[0]
                return loss if alpha_t is None else loss * alpha_t
            y = align.forward_tokens(x_student) if align is not None else x_student
            return crit.forward_tokens(y, x_teacher, gt_semantic_seg, step)
        x_s, x_t = _to_nchw(x_student), _to_nchw(x_teacher)
        if align is not None:
            x_s = align(x_s)
        return crit(x_s, x_t, gt_semantic_seg, step)

    def _fuse_pairs(self, losses, student_features, teacher_features, gt_semantic_seg, step):
        """Two channel criteria on the SAME pair of taps (BASELINE config 3: CGD g = 8 + channel-wise KL g = 1 on decode_head.linear_pred;
        SURVEY section 7 step 7) run as ONE fused-upsample pass each way (ops.cgd_kl_up2) instead of two passes and a gradient add.  Fused only
        when one permutation table can order the channel slots of both: the second criterion has no shuffle of its own and either a group
        size of 1 (slot order immaterial) or a partner without a shuffle as well."""
        import os
        from .losses import KLDLoss
        if os.environ.get('SEGDISTILL_FUSE_PAIRS', '1') != '1':        # A/B: every criterion as its own pass
            return
        by_taps = {}
        for i, entry in enumerate(self.distillation):
            if losses[i] is not None or str(i) in self.aligns or not isinstance(self.criteria[i], KLDLoss) or isinstance(entry['student_layer'], list):
                continue
            by_taps.setdefault((entry['student_layer'], entry['teacher_layer']), []).append(i)
        for (s_name, t_name), idx in by_taps.items():
            xs, xt = student_features[s_name], teacher_features[t_name]
            while len(idx) >= 2:
                ia = idx.pop(0)
                ca = self.criteria[ia]
                size = ca.fused_up_size(xs, xt, gt_semantic_seg) if xs.dim() == 4 and xt.dim() == 4 else None
                partner = None
                for ib in idx:
                    cb = self.criteria[ib]
                    if size is None or cb.fused_up_size(xs, xt, gt_semantic_seg) != size:
                        continue
                    # the criterion WITH a shuffle (if any) leads; the other one must not care about the slot order
                    lead, other = (ca, cb) if (ca.shuffle_config or not cb.shuffle_config) else (cb, ca)
                    if other.shuffle_config or (lead.shuffle_config and other.transform_config['group_size'] != 1):
                        continue
                    partner = ib
                    break
                if partner is None:
                    continue
                idx.remove(partner)
                cb = self.criteria[partner]
                first, second = (ia, partner) if (ca.shuffle_config or not cb.shuffle_config) else (partner, ia)
                cf, cs_ = self.criteria[first], self.criteria[second]
                alpha_f, at_f, perm = cf._prepare(xs, xs.shape[1], step)
                alpha_s, at_s, _ = cs_._prepare(xs, xs.shape[1], step)
                lf, ls = ops.cgd_kl_up2(xs, xt, size, (cf.transform_config['group_size'], cf.tau, alpha_f),
                                        (cs_.transform_config['group_size'], cs_.tau, alpha_s), perm)
                losses[first] = lf if at_f is None else lf * at_f
                losses[second] = ls if at_s is None else ls * at_s

    def forward(self, student_features, teacher_features, gt_semantic_seg, step, student=None, teacher=None):
        losses = [None] * len(self.distillation)
        # token-major entries first: their align projections run one by one, their criteria in ONE call each way (config 5 taps four decoder
        # stages; evaluated one after the other -- reference opts.py:100-110 -- each criterion was four chained launches forward)
        batch, fused = [], []
        for i, entry in enumerate(self.distillation):
            s_name, t_name = entry['student_layer'], entry['teacher_layer']
            if isinstance(s_name, list):
                continue                    # attention-pair entry (reference opts.py:91-98): dispatched in the naming loop below
            xs, xt = student_features[s_name], teacher_features[t_name]
            align = self.aligns[str(i)] if str(i) in self.aligns else None
            crit = self.criteria[i]
            if hasattr(crit, 'host_prepare') and not (xs.is_cuda and torch.cuda.is_current_stream_capturing()):
                # schedules and shuffle draws in ENTRY order whatever the launch order below (the draws come from one CPU generator)
                crit.host_prepare(step, align.weight.shape[0] if align is not None else xs.shape[2 if xs.dim() == 3 else 1])
            if self._token_form(i, xs, xt):
                if align is not None and ops.align_cgd_tokens_supported(xs, align.weight, xt):
                    fused.append((i, xs, align, xt))
                else:
                    batch.append((i, align.forward_tokens(xs) if align is not None else xs, xt))
        # entries whose projection feeds nothing but the criterion: projection + criterion fused (csrc/align_tok.hip), all stages of one student
        # width in ONE scan launch + ONE finish launch forward and one launch backward
        by_k = {}
        for item in fused:
            by_k.setdefault(item[1].shape[2], []).append(item)
        cap = ops.cgd_kl_tokens_max_jobs() if fused else 1
        for items in by_k.values():
            for lo in range(0, len(items), cap):
                part = items[lo:lo + cap]
                jobs = [self.criteria[i].token_job(xt, step) for i, _, _, xt in part]
                out = ops.align_cgd_tokens_multi([(xs, al.weight, al.bias, xt) for _, xs, al, xt in part], [m for m, _ in jobs])
                for (i, _, _, _), (_, alpha_t), loss in zip(part, jobs, out):
                    losses[i] = loss if alpha_t is None else loss * alpha_t
        by_dtype = {}
        for item in batch:
            by_dtype.setdefault(item[1].dtype, []).append(item)
        cap = ops.cgd_kl_tokens_max_jobs() if batch else 1
        for items in by_dtype.values():
            for lo in range(0, len(items), cap):
                part = items[lo:lo + cap]
                jobs = [self.criteria[i].token_job(y, step) for i, y, _ in part]
                out = ops.cgd_kl_tokens_multi([(y, xt) for _, y, xt in part], [m for m, _ in jobs])
                for (i, _, _), (_, alpha_t), loss in zip(part, jobs, out):
                    losses[i] = loss if alpha_t is None else loss * alpha_t
        self._fuse_pairs(losses, student_features, teacher_features, gt_semantic_seg, step)
        out = {}
        for i, entry in enumerate(self.distillation):
            s_name, t_name = entry['student_layer'], entry['teacher_layer']
            if isinstance(s_name, list):
                # reference opts.py:91-98: two taps per network (attention map + value) and the networks themselves go to a criterion with the
                # 8-argument signature; none of the reference's LIVE loss classes has it (they sit in the commented-out generation of
                # losses.py), so this only serves criteria a user registers in DISTILL_LOSSES.  Key as the reference names it.
                loss = self.criteria[i](student_features[s_name[0]], student_features[s_name[1]], teacher_features[t_name[0]],
                                        teacher_features[t_name[1]], student, teacher, gt_semantic_seg, step)
                out[f"loss_{s_name[0]}<->{t_name}_{entry['loss_name']}"] = loss
                continue
            loss = losses[i]
            if loss is None:
                loss = self.entry_loss(i, student_features[s_name], teacher_features[t_name], gt_semantic_seg, step)
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

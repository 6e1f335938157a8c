"""Distillation criteria with the reference's names, constructor kwargs and call signature
(``criterion(x_student, x_teacher, gt_semantic_seg, n_iter) -> 0-dim tensor``), computed by
the HIP kernels of libsegdistill_hip.so.

Interface counterpart of reference mmseg/models/distillation/losses.py: KLDLoss :9-113,
PDLoss :115-128, CDLoss :130-143, CGDLoss :145-158, CGDLossWS :160-173, ATLoss :175-197,
IFVDLoss :199-238.  Host-side decisions (alpha schedule :61-92, whether this is a shuffle
iteration :38) are taken here; everything per-pixel runs on the GPU:

* 'channel' rows (CD / CGD): ``ops.cgd_kl`` -- grouped-channel online softmax + KL, the -1e9 pad
  of :55-58 is virtual, the channel shuffle of :39-41 is a permutation table read by the
  kernel (no gather copies);
* 'pixel' rows (PD): ``ops.pix_kl``.

On a CPU tensor the ops raise; the one exception is the explicit plumbing mode below (BASELINE configs[0]), which is never a fallback.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..builder import DISTILL_LOSSES


# BASELINE configs[0] ("CPU train_step via tools/train.py (plumbing, no GPU)"): an EXPLICIT opt-in (tools/train.py --cpu-plumbing, or
# SEGDISTILL_CPU_PLUMBING=1) under which KLDLoss evaluates CPU taps with ATen ops, so that the whole harness -- config loader, registries, hooks,
# loss naming, optimizer, checkpointing -- can be exercised on a box without a GPU.  It is never taken for CUDA tensors and never by default:
# without the switch a CPU tap raises (tests/test_host_logic_cpu.py), and on a GPU box a missing extension still fails loudly (_lib.lib()).
CPU_PLUMBING = os.environ.get('SEGDISTILL_CPU_PLUMBING', '0') == '1'


def _plumbing_kld(s, t, gt, alpha, tau, resize_config, transform_config, perm):
    """reference losses.py:101-112 on CPU tensors, op for op (resize both, gather the shuffle, -1e9 pad, rows, KLDivLoss(sum) / rows * alpha)."""
    if resize_config:
        size = tuple(int(v) for v in (t if resize_config.get('target', 'gt') == 'teacher' else gt).shape[2:])
        s, t = (F.interpolate(v, size=size, mode=resize_config['mode'], align_corners=resize_config['align_corners']) for v in (s, t))
    if perm is not None:
        s, t = s[:, perm.long()].contiguous(), t[:, perm.long()].contiguous()
    kind = transform_config['loss_type'] if transform_config else None
    if kind == 'pixel':
        s, t = (v.permute(0, 2, 3, 1).flatten(1, 2) for v in (s, t))
    elif kind == 'channel':
        g = transform_config['group_size']
        n = (g - s.shape[1] % g) % g
        if n:
            s, t = (torch.cat([v, v.new_full((v.shape[0], n) + tuple(v.shape[2:]), -1e9)], 1) for v in (s, t))
        s, t = (v.reshape(v.shape[0], v.shape[1] // g, -1) for v in (s, t))
    elif kind is not None:
        raise ValueError(f'unknown loss_type {kind!r}')
    kl = F.kl_div(F.log_softmax(s / tau, -1), F.softmax(t / tau, -1), reduction='sum')
    return kl / (s.numel() // s.shape[-1]) * alpha


def _bilinear(x, size):
    from ..layers import resize            # csrc/resize.hip for contiguous NCHW maps on the GPU, ATen otherwise
    return resize(x, size=tuple(int(v) for v in size), mode='bilinear', align_corners=False, warning=False, alias_ok=True)


@DISTILL_LOSSES.register_module()
class KLDLoss(nn.Module):
    def __init__(self, alpha=1, tau=1, resize_config=None, shuffle_config=None, transform_config=None, warmup_config=None,
                 earlydecay_config=None):
        super().__init__()
        self.alpha_0 = alpha
        self.alpha = alpha
        self.tau = tau
        self.resize_config = resize_config
        self.shuffle_config = shuffle_config
        self.transform_config = transform_config
        self.warmup_config = warmup_config
        self.earlydecay_config = earlydecay_config
        self.fuse_resize = True   # use the fused-upsample kernels when the resize is bilinear/align_corners=False
        self.last_perm = None     # permutation used by the most recent shuffle iteration (for tests / logging)
        # hipGraph support: host-side decisions (alpha schedule, shuffle draw) are made by host_prepare(n_iter) and
        # reach the kernels as DATA (a 1-element alpha tensor, a permutation table that is the identity on
        # non-shuffle iterations), so that a captured step can be replayed while they change.
        self.graph_safe = False
        self._prepared_for = None
        self._perm_host = None
        self._alpha_t = None
        self._perm_t = None
        self._alpha_synced = None
        self._perm_synced = 'unset'

    # ---- host-side schedule: same state machine as reference losses.py:61-92 -------------------
    def warmup(self, n_iter):
        cfg = self.warmup_config
        total = cfg['warmup_iters']
        if n_iter > total:
            return
        if n_iter == total:
            self.alpha = self.alpha_0
        elif cfg['mode'] == 'linear':
            self.alpha = self.alpha_0 * (n_iter / total)
        elif cfg['mode'] == 'exp':
            self.alpha = self.alpha_0 ** (n_iter / total)
        elif cfg['mode'] == 'jump':
            self.alpha = 0

    def earlydecay(self, n_iter):
        cfg = self.earlydecay_config
        start, end = cfg['earlydecay_start'], cfg['earlydecay_end']
        if n_iter < start:
            return
        if start < n_iter < end:
            left = (end - n_iter) / (end - start)
            if cfg['mode'] == 'linear':
                self.alpha = self.alpha_0 * left
            elif cfg['mode'] == 'exp':
                self.alpha = 0.001 * self.alpha_0 ** left
            elif cfg['mode'] == 'jump':
                self.alpha = 0
        elif n_iter >= end:
            self.alpha = 0

    def _draw_perm_host(self, channels, n_iter):
        """reference :38-41 draws torch.randperm(C) from the CPU global RNG on every rank independently;
        ranks are made to agree by broadcasting rank 0's draw."""
        if not self.shuffle_config or n_iter % self.shuffle_config['interval'] != 0:
            return None
        perm = torch.randperm(channels)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == 'nccl':
                p = perm.cuda()
                dist.broadcast(p, src=0)
                perm = p.cpu()
            else:
                dist.broadcast(perm, src=0)
        self.last_perm = perm
        return perm

    def host_prepare(self, n_iter, channels):
        """All host-side per-iteration decisions of reference :96-99 and :38 (idempotent per n_iter)."""
        if self._prepared_for == n_iter:
            return
        if self.warmup_config:
            self.warmup(n_iter)
        if self.earlydecay_config:
            self.earlydecay(n_iter)
        self._perm_host = self._draw_perm_host(channels, n_iter) if self.shuffle_config else None
        self._prepared_for = n_iter

    def sync_device_state(self, channels, device):
        """graph_safe mode: push alpha / permutation into the static device buffers the captured kernels read
        (only when they changed: a steady-state step issues no copy at all)."""
        if self._alpha_t is None or self._alpha_t.device != device:
            self._alpha_t = torch.empty((), dtype=torch.float32, device=device)
            self._alpha_synced = None
        if self._alpha_synced != float(self.alpha):
            self._alpha_t.fill_(float(self.alpha))
            self._alpha_synced = float(self.alpha)
        if self.shuffle_config:
            if self._perm_t is None or self._perm_t.numel() != channels or self._perm_t.device != device:
                self._perm_t = torch.empty(channels, dtype=torch.int32, device=device)
                self._perm_synced = 'unset'
            want = None if self._perm_host is None else tuple(self._perm_host.tolist())
            if self._perm_synced != want:
                src = torch.arange(channels) if want is None else self._perm_host
                self._perm_t.copy_(src.to(torch.int32))
                self._perm_synced = want

    def prepare_replay(self, n_iter):
        """Called by the trainer before replaying a captured step: same host work as forward() would do."""
        channels, device = self._seen
        self.host_prepare(n_iter, channels)
        self.sync_device_state(channels, device)

    # ---- token-major taps [B, N, C] (SegFormer decoder features; SURVEY a-16) ----------------------------------------------------
    def tokens_ok(self, x_teacher):
        """This criterion can run on token-major operands as they are: 'channel' rows, no resize (a token sequence has no (h, w) to resize
        to the label size; config 5 uses resize_config=None), whole 16-byte channel vectors."""
        if self.resize_config or not self.transform_config or self.transform_config.get('loss_type') != 'channel':
            return False
        n = 4 if x_teacher.dtype == torch.float32 else 8
        return x_teacher.is_cuda and x_teacher.dtype in (torch.float32, torch.bfloat16) and x_teacher.shape[2] % n == 0 and \
            (not self.shuffle_config or x_teacher.shape[2] <= 2048)

    def forward_tokens(self, x_student, x_teacher, gt, n_iter):
        """forward() for token-major operands [B, N, C] (same schedule / shuffle state machine, csrc/cgd_tok.hip kernels)."""
        return self.forward(x_student, x_teacher, gt, n_iter, _tokens=True)

    def _prepare(self, x_student, channels, n_iter):
        """Host-side decisions of this call (schedule, shuffle draw) -> (alpha for the kernel, device alpha factor or None, perm or None)."""
        self._seen = (channels, x_student.device)
        capturing = x_student.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.host_prepare(n_iter, channels)
            if self.graph_safe:
                self.sync_device_state(channels, x_student.device)
        if self.graph_safe:
            return 1.0, self._alpha_t, (self._perm_t if self.shuffle_config else None)
        perm = None if self._perm_host is None else self._perm_host.to(device=x_student.device, dtype=torch.int32)
        return self.alpha, None, perm

    def token_job(self, x_student, n_iter):
        """The token-major form of this call as a job of ops.cgd_kl_tokens_multi: ((group_size, tau, alpha, perm), alpha_t) -- the caller
        batches the jobs of several entries into one launch each way and multiplies loss_i by alpha_t_i when it is not None."""
        alpha, alpha_t, perm = self._prepare(x_student, x_student.shape[2], n_iter)
        return (self.transform_config['group_size'], self.tau, alpha, perm), alpha_t

    def forward(self, x_student, x_teacher, gt, n_iter, _tokens=False):
        if _tokens:
            meta, alpha_t = self.token_job(x_student, n_iter)
            loss = ops.cgd_kl_tokens_multi([(x_student, x_teacher)], [meta])[0]
            return loss if alpha_t is None else loss * alpha_t
        alpha, alpha_t, perm = self._prepare(x_student, x_student.shape[1], n_iter)
        if CPU_PLUMBING and not x_student.is_cuda and not x_teacher.is_cuda and x_student.dim() == 4:
            return _plumbing_kld(x_student, x_teacher, gt, alpha, self.tau, self.resize_config, self.transform_config, perm)
        loss = self._device_part(x_student, x_teacher, gt, alpha, perm)
        return loss if alpha_t is None else loss * alpha_t

    def fused_up_size(self, x_student, x_teacher, gt):
        """The label size this call would up-sample both taps to INSIDE the fused R2 kernels (ops.cgd_kl_up), or None when it takes another
        route -- what DistillationLoss needs to know to run two such criteria on the same taps as one pass each way (ops.cgd_kl_up2)."""
        if not self.resize_config or not self.transform_config or self.transform_config.get('loss_type') != 'channel' or not self.fuse_resize:
            return None
        if x_student.dim() != 4 or self.resize_config['mode'] != 'bilinear' or self.resize_config['align_corners']:
            return None
        ref = x_teacher if self.resize_config.get('target', 'gt') == 'teacher' else gt
        out_size = tuple(int(v) for v in ref.shape[2:])
        return out_size if ops.can_fuse_resize(x_student, x_teacher, out_size, self.transform_config) else None

    def _device_part(self, x_student, x_teacher, gt, alpha, perm):
        out_size = None
        if self.resize_config:
            # reference :101-102 resizes BOTH tensors to the label size; 'target': 'teacher' (an extension used
            # by feature-level configs) resizes to the teacher tap's size instead.
            ref = x_teacher if self.resize_config.get('target', 'gt') == 'teacher' else gt
            out_size = tuple(int(v) for v in ref.shape[2:])
            mode, ac = self.resize_config['mode'], self.resize_config['align_corners']
            fusable = mode == 'bilinear' and not ac and self.fuse_resize and \
                ops.can_fuse_resize(x_student, x_teacher, out_size, self.transform_config)
            if (not fusable and mode == 'bilinear' and not ac and self.fuse_resize and self.transform_config
                    and self.transform_config.get('loss_type') == 'pixel' and ops.can_fuse_pixel_resize(x_student, x_teacher, out_size)):
                # PDLoss (reference :115-128): class softmax at the label size straight from the taps (csrc/pix_up.hip).  A shuffle (:39-41)
                # permutes the classes of every pixel alike and leaves a softmax over all of them unchanged: no gather
                return ops.pix_kl_up(x_student, x_teacher, out_size, tau=self.tau, alpha=alpha)
            if not fusable:
                from ..layers import resize        # csrc/resize.hip for contiguous NCHW maps on the GPU, ATen otherwise
                if tuple(x_student.shape[2:]) != out_size:
                    x_student = resize(x_student, size=out_size, mode=mode, align_corners=ac, warning=False, alias_ok=True)
                if tuple(x_teacher.shape[2:]) != out_size:
                    x_teacher = resize(x_teacher, size=out_size, mode=mode, align_corners=ac, warning=False, alias_ok=True)
                out_size = None
        kind = self.transform_config['loss_type'] if self.transform_config else None
        if kind == 'channel':
            g = self.transform_config['group_size']
            if out_size is not None:
                return ops.cgd_kl_up(x_student, x_teacher, out_size, group_size=g, tau=self.tau, alpha=alpha, perm=perm)
            return ops.cgd_kl(x_student, x_teacher, group_size=g, tau=self.tau, alpha=alpha, perm=perm)
        if perm is not None:  # the remaining row layouts have no permutation-table kernel: gather as the reference does
            idx = perm.long()
            x_student, x_teacher = x_student[:, idx].contiguous(), x_teacher[:, idx].contiguous()
        if kind == 'pixel':
            if out_size is not None:
                x_student, x_teacher = _bilinear(x_student, out_size), _bilinear(x_teacher, out_size)
            return ops.pix_kl(x_student, x_teacher, tau=self.tau, alpha=alpha)
        if kind is None:
            # reference :108-111 on an untransformed 4-D tensor: softmax over the LAST axis,
            # rows = B*C*H.  Same kernel, seen as B*C*H single-plane rows of W elements.
            if out_size is not None:
                x_student, x_teacher = _bilinear(x_student, out_size), _bilinear(x_teacher, out_size)
            w = x_student.shape[-1]
            s2 = x_student.reshape(1, -1, 1, w)
            t2 = x_teacher.reshape(1, -1, 1, w)
            return ops.cgd_kl(s2, t2, group_size=1, tau=self.tau, alpha=alpha)
        raise ValueError(f'unknown loss_type {kind!r}')


_BILINEAR = {'mode': 'bilinear', 'align_corners': False}


@DISTILL_LOSSES.register_module()
class PDLoss(KLDLoss):
    """Pixel-wise distillation: softmax over classes at every pixel (reference :115-128)."""

    def __init__(self):
        super().__init__(alpha=1, tau=1, resize_config=dict(_BILINEAR), transform_config={'loss_type': 'pixel'})


@DISTILL_LOSSES.register_module()
class CDLoss(KLDLoss):
    """Channel-wise distillation: softmax over the H*W pixels of every channel (reference :130-143)."""

    def __init__(self):
        super().__init__(alpha=1, tau=1, resize_config=dict(_BILINEAR), transform_config={'loss_type': 'channel', 'group_size': 1})


@DISTILL_LOSSES.register_module()
class CGDLoss(KLDLoss):
    """Channel Group Distillation (reference :145-158): softmax over group_size channels x H*W,
    channel shuffle every 1000 iterations."""

    def __init__(self, group_size=10, alpha=3, tau=2):
        super().__init__(alpha=alpha, tau=tau, resize_config=dict(_BILINEAR), shuffle_config={'interval': 1000},
                         transform_config={'loss_type': 'channel', 'group_size': group_size})


@DISTILL_LOSSES.register_module()
class CGDLossWS(KLDLoss):
    """CGD with linear warm-up (2000 it) and linear early decay 110k -> 120k (reference :160-173)."""

    def __init__(self):
        super().__init__(alpha=3, tau=2, resize_config=dict(_BILINEAR), shuffle_config={'interval': 1000},
                         transform_config={'loss_type': 'channel', 'group_size': 10},
                         warmup_config={'mode': 'linear', 'warmup_iters': 2000},
                         earlydecay_config={'mode': 'linear', 'earlydecay_start': 110000, 'earlydecay_end': 120000})


@DISTILL_LOSSES.register_module()
class ATLoss(nn.Module):
    """Attention transfer + pixel KL (reference :175-197): MSE between channel-mean maps plus
    the class-softmax KL (tau=1, alpha=1) of the un-resized logits."""

    def forward(self, x_student, x_teacher, gt, step):
        return ops.at_kl(x_student, x_teacher)      # both terms in one pass over the logits each way (csrc/pix_kl.hip, AT form)


@DISTILL_LOSSES.register_module()
class IFVDLoss(nn.Module):
    """Intra-class feature variation distillation (reference :199-238): cosine similarity of
    every pixel to its class centre, matched between student and teacher (x10 MSE), plus the
    class-softmax KL.  The reference's 150-pass mask loop (:226-230) is one segmented mean over
    class-sorted pixels (csrc/ifvd.hip).  GPU only, like every criterion of this module."""


    def forward(self, preds_S, preds_T, target, step):
        feat_T = _bilinear(preds_T, preds_S.shape[2:])
        n_cls = feat_T.shape[1]
        pd = ops.pix_kl(preds_S, feat_T, tau=1.0, alpha=1.0)
        lab = F.interpolate(target.float(), size=preds_S.shape[2:], mode='nearest')
        # csrc/ifvd.hip: class means as sorted-run gathers, cosine pass, gradient through the centres -- no mask loop, no atomics
        cls = torch.where((lab >= 0) & (lab < n_cls) & (lab == lab.floor()), lab, torch.full_like(lab, -1.)).to(torch.int32)
        return ops.ifvd_term(preds_S, feat_T.detach(), cls, n_cls) + pd

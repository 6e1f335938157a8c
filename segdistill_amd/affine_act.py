"""Eval-mode BatchNorm (+ residual) (+ ReLU) of a frozen convolutional network in one in-place pass (csrc/affine_act.hip).

reference: `relu(bn(conv(x)))`, `out += identity; relu(out)` of backbones/resnet.py:18-100 and ConvModule's conv -> norm -> act (psp_head.py:38-44,
84-91; uper_head.py:30-75) for a teacher in eval mode: MIOpen's inference BatchNorm, an add and a ReLU are three launches and three passes."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from . import _lib
from .layers import frozen_derived
from .ops import _stream_ptr

_ENABLED = os.environ.get('SEGDISTILL_FUSED_EVAL_BN', '1') == '1'      # A/B: 0 = torch's batch_norm / add / relu launches


def usable(x, norm):
    """x: a fresh contiguous NCHW fp32 map on the GPU nobody differentiates through; norm: a BatchNorm in eval mode with running statistics."""
    return (_ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and not torch.is_grad_enabled()
            and isinstance(norm, _BatchNorm) and not norm.training and norm.track_running_stats and norm.running_mean is not None and norm.affine
            and not norm.weight.requires_grad and not (norm._forward_hooks or norm._forward_pre_hooks) and x.numel() > 0)


def eval_norm_act_(x, norm, relu, residual=None):
    """x <- act(norm(x) (+ residual)), IN PLACE; returns x.  The per-channel scale / shift are cached while the norm's tensors are unchanged."""
    scale = frozen_derived(norm.weight, 'bn_scale', lambda: (norm.weight * torch.rsqrt(norm.running_var + norm.eps)).float(), norm.running_var)
    shift = frozen_derived(norm.bias, 'bn_shift', lambda: (norm.bias - norm.running_mean * scale).float(), norm.running_mean, norm.running_var,
                           norm.weight)
    B, C, H, W = x.shape
    r = None
    if residual is not None:
        r = residual if (residual.is_contiguous() and residual.dtype == torch.float32) else residual.float().contiguous()
        assert r.shape == x.shape
    _lib.check(_lib.lib().sd_affine_act_nchw(x.data_ptr(), None if r is None else r.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                             B * C, C, H * W, int(bool(relu)), _stream_ptr()), 'sd_affine_act_nchw')
    return x


def run_frozen_sequential(seq, x):
    """nn.Sequential of conv / norm / ReLU / pooling layers (a ResNet stem or shortcut): every `conv -> BatchNorm (-> ReLU)` run is fused."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(m, nn.Conv2d) and isinstance(nxt, _BatchNorm) and not (m._forward_hooks or nxt._forward_hooks):
            y = m(x)
            act = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU) and not mods[i + 2]._forward_hooks
            if usable(y, nxt):
                x = eval_norm_act_(y, nxt, act)
                i += 3 if act else 2
                continue
            x = nxt(y)
            i += 2
            continue
        x = m(x)
        i += 1
    return x

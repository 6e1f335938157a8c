"""Building blocks the reference takes from mmcv-full 1.2.2 / timm 0.3.2 (neither is
vendored in the reference nor installed here), restated with the same child-module names
so that reference checkpoints' state-dict keys line up:

* ``ConvModule``        conv -> norm -> act; children ``conv``, ``bn``/``gn``, ``activate``;
                        conv bias defaults to ``norm_cfg is None``; kaiming-normal
                        (fan_out, relu) conv init, norm weight 1 / bias 0.
* ``build_norm_layer``  returns ``(abbr+postfix, layer)`` with abbr bn/gn/ln.
* ``DropPath``          per-sample stochastic depth.
* ``resize``            F.interpolate wrapper (reference mmseg/ops/wrappers.py:8-29).
"""
from __future__ import annotations

import os
import warnings
import weakref

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F


def _want_sync_bn():
    """A multi-rank GPU process group exists.  SEGDISTILL_FORCE_SYNCBN=1 also builds the synchronised layer for a ONE-rank
    group, so that its RCCL collectives (and the segmented graph capture around them) can be exercised on a one-GPU box."""
    if not (dist.is_available() and dist.is_initialized() and torch.cuda.is_available()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('SEGDISTILL_FORCE_SYNCBN') == '1'


class ChainedSyncBatchNorm(nn.SyncBatchNorm):
    """torch.nn.SyncBatchNorm (same parameters, buffers, state-dict keys, arithmetic).  While a trainer records a
    SEGMENTED hipGraph step it points ``_segments`` at its recorder and the training-mode forward goes through
    ``SegmentRecorder.sync_batch_norm``: the same ATen sequence with the two collectives exposed as cut points between
    graphs (engine/segments.py).  Otherwise this is exactly the parent class."""

    _segments = None

    def forward(self, x):
        rec = self._segments
        if rec is None or not self.training or not x.is_cuda:
            return super().forward(x)
        return rec.sync_batch_norm(self, x)


def build_norm_layer(cfg, num_features, postfix=''):
    """``cfg``: dict(type='BN'|'SyncBN'|'GN'|'LN', requires_grad=True, eps=..., ...).

    'SyncBN' builds a torch.nn.SyncBatchNorm (ChainedSyncBatchNorm) when a multi-rank GPU process group exists and
    plain BatchNorm2d otherwise (identical math on one rank, identical state-dict keys).
    """
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise KeyError('the norm cfg must be a dict containing the key "type"')
    opts = dict(cfg)
    kind = opts.pop('type')
    trainable = opts.pop('requires_grad', True)
    opts.setdefault('eps', 1e-5)
    if kind in ('BN', 'BN2d'):
        abbr, layer = 'bn', nn.BatchNorm2d(num_features, **opts)
    elif kind == 'SyncBN':
        abbr = 'bn'
        layer = ChainedSyncBatchNorm(num_features, **opts) if _want_sync_bn() else nn.BatchNorm2d(num_features, **opts)
    elif kind == 'GN':
        abbr, layer = 'gn', nn.GroupNorm(num_channels=num_features, **opts)
    elif kind == 'LN':
        abbr, layer = 'ln', nn.LayerNorm(num_features, **opts)
    else:
        raise KeyError(f'Unrecognized norm type {kind}')
    for p in layer.parameters():
        p.requires_grad = trainable
    return f'{abbr}{postfix}', layer


def build_conv_layer(cfg, *args, **kwargs):
    kind = 'Conv2d' if cfg is None else cfg.get('type', 'Conv2d')
    if kind not in ('Conv2d', 'Conv'):
        raise KeyError(f'Unrecognized conv type {kind} (DCN and friends are outside the KD path)')
    return nn.Conv2d(*args, **kwargs)


def kaiming_init(m, mode='fan_out', nonlinearity='relu', bias=0.):
    if getattr(m, 'weight', None) is not None:
        nn.init.kaiming_normal_(m.weight, a=0, mode=mode, nonlinearity=nonlinearity)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def constant_init(m, val, bias=0.):
    if getattr(m, 'weight', None) is not None:
        nn.init.constant_(m.weight, val)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def normal_init(m, mean=0., std=1., bias=0.):
    if getattr(m, 'weight', None) is not None:
        nn.init.normal_(m.weight, mean, std)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        use_bias = (not self.with_norm) if bias == 'auto' else bool(bias)
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                     dilation=dilation, groups=groups, bias=use_bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm_name = None
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            if act_cfg.get('type') != 'ReLU':
                raise KeyError(f'activation {act_cfg} is not used on the KD path')
            self.activate = nn.ReLU(inplace=inplace)
        kaiming_init(self.conv)
        if self.with_norm:
            constant_init(self.norm, 1, bias=0)

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.norm_name else None

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            if not torch.is_grad_enabled() and x.is_cuda and not (self.with_activation and (self.activate._forward_hooks or self.activate._forward_pre_hooks)):
                from . import affine_act
                if affine_act.usable(x, self.norm):
                    # frozen network in eval mode: BatchNorm + ReLU as one in-place pass over the conv's output (csrc/affine_act.hip)
                    return affine_act.eval_norm_act_(x, self.norm, self.with_activation)
            x = self.norm(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class DropPath(nn.Module):
    """Stochastic depth per sample: train -> x/(1-p) * Bernoulli(1-p); eval -> identity."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.training or self.drop_prob == 0.:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask.div_(keep)   # same value as x/keep * mask with one pass over x instead of two

    def sample_scale(self, x):
        """The per-sample factor [B] (fp32) this module would multiply `x` by, or None when it is the identity -- for callers
        that fuse the multiplication into another kernel (layernorm.add_layernorm)."""
        if not self.training or self.drop_prob == 0.:
            return None
        keep = 1.0 - self.drop_prob
        return torch.empty(x.shape[0], dtype=torch.float32, device=x.device).bernoulli_(keep).div_(keep)

    def extra_repr(self):
        return f'p={self.drop_prob}'


def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)


def to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def resize(input, size=None, scale_factor=None, mode='nearest', align_corners=None, warning=True, alias_ok=False):
    """The reference's `resize` wrapper (mmseg/ops/wrappers.py:8-28).  alias_ok: the caller only READS the result, so a same-size request may
    return the input itself (the distillation criteria); everybody else gets a fresh tensor, as F.interpolate gives (UPerHead / FPN
    style `+=` or relu_ on an aliased result would have written into the source feature, possibly a tapped one)."""
    if warning and size is not None and align_corners:
        ih, iw = (int(v) for v in input.shape[2:])
        oh, ow = (int(v) for v in size)
        if (oh > ih or ow > iw) and min(oh, ow, ih, iw) > 1 and (oh - 1) % (ih - 1) and (ow - 1) % (iw - 1):
            warnings.warn(f'align_corners={align_corners}: sizes {(ih, iw)} -> {(oh, ow)} are not of the form x+1 -> nx+1')
    if isinstance(size, torch.Size):
        size = tuple(int(v) for v in size)
    if scale_factor is None and size is not None:
        from . import resize as hip_resize
        if hip_resize.supported(input, size, mode, align_corners):
            if tuple(input.shape[2:]) == tuple(int(v) for v in size):
                return input if alias_ok else input.clone()
            return hip_resize.bilinear(input, size, bool(align_corners))     # csrc/resize.hip: contiguous NCHW maps on the GPU
    return F.interpolate(input, size, scale_factor, mode, align_corners)


def add_prefix(d, prefix):
    """reference mmseg/core/utils/misc.py:1 -- names the decode./aux. loss keys."""
    return {f'{prefix}.{k}': v for k, v in d.items()}


_FROZEN = {}   # id(frozen parameter) -> (weak reference to it, {key: (versions, derived tensor)})


def _frozen_slot(param):
    ent = _FROZEN.get(id(param))
    if ent is None or ent[0]() is not param:            # first use, or the id was recycled after the parameter died
        ent = (weakref.ref(param, lambda _, k=id(param): _FROZEN.pop(k, None)), {})
        _FROZEN[id(param)] = ent
    return ent[1]


def frozen_derived(param, key, fn, *also):
    """`fn()` -- a re-layout / product derived from `param` (and the tensors in `also`) -- cached for FROZEN parameters
    (requires_grad False: the teacher), keyed on the in-place version counters so a later checkpoint load invalidates it.
    A trainable parameter changes every step and is never cached.  Nothing is stored while a hipGraph is being captured
    (a tensor allocated inside a capture lives in the graph's private pool)."""
    if param.requires_grad or any(t.requires_grad for t in also):
        return fn()
    versions = (param._version, param.data_ptr()) + tuple((t._version, t.data_ptr()) for t in also)
    slot = _frozen_slot(param)
    hit = slot.get(key)
    if hit is not None and hit[0] == versions:
        return hit[1]                   # also while capturing: an entry made by the eager warm-up passes is an ordinary static tensor
    if param.is_cuda and torch.cuda.is_current_stream_capturing():
        return fn()                     # computed inside the capture (a node of the graph), never stored
    with torch.no_grad():
        value = fn()
    slot[key] = (versions, value)
    return value


def tokens_of(x):
    """[B,C,H,W] -> token-major [B, H*W, C].  For a channels-last tensor (what MIOpen convs return for channels-last
    input, and what ``nchw_view_of_tokens`` produces) this is a VIEW; for a contiguous NCHW tensor it is the usual
    flatten(2).transpose(1,2) (a strided view that the next op materialises)."""
    b, c, h, w = x.shape
    if x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        return x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    return x.flatten(2).transpose(1, 2)


def nchw_view_of_tokens(tokens, hw):
    """token-major [B, H*W, C] -> [B,C,H,W] WITHOUT a copy: a logically-NCHW tensor with channels-last strides."""
    b, n, c = tokens.shape
    return tokens.reshape(b, hw[0], hw[1], c).permute(0, 3, 1, 2)

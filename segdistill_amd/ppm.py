"""All adaptive average pools of a Pyramid Pooling Module in one pass each way (csrc/ppm_pool.hip; reference
mmseg/models/decode_heads/psp_head.py:10-58): forward reads the map once and writes every scale's [s x s] map, backward GATHERS dx from the
pooled gradients -- deterministic, no float atomics, and the branch gradients are never materialised as full maps."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from .ops import _DT, _stream_ptr

ENABLED = os.environ.get('SEGDISTILL_PPM_POOL', '1') == '1'      # A/B: 0 = nn.AdaptiveAvgPool2d per scale (ATen, atomic backward)


def _scales_arr(scales):
    return (C.c_int * len(scales))(*[int(s) for s in scales])


def supported(x, scales):
    if not (ENABLED and x.is_cuda and x.dim() == 4 and x.dtype in _DT and x.is_contiguous() and x.numel() > 0 and 0 < len(scales) <= 4):
        return False
    if not all(isinstance(s, int) for s in scales):
        return False
    return bool(_lib.lib().sd_ppm_pool_supported(int(x.shape[2]), int(x.shape[3]), _scales_arr(scales), len(scales)))


class _PPMPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scales):
        B, Cc, h, w = x.shape
        outs = [torch.empty(B, Cc, s, s, dtype=x.dtype, device=x.device) for s in scales]
        ptrs = (C.c_void_p * len(scales))(*[o.data_ptr() for o in outs])
        _lib.check(_lib.lib().sd_ppm_pool_fwd(x.data_ptr(), _DT[x.dtype], B * Cc, h, w, _scales_arr(scales), len(scales), ptrs, _stream_ptr()),
                   'sd_ppm_pool_fwd')
        ctx.geom = (B, Cc, h, w, tuple(scales), x.dtype)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        B, Cc, h, w, scales, dt = ctx.geom
        dev = next(g.device for g in grads if g is not None)       # autograd calls backward only when at least one output has a gradient
        gs = []
        for g, s in zip(grads, scales):
            g = torch.zeros(B, Cc, s, s, dtype=dt, device=dev) if g is None else g
            gs.append(g.to(dt).contiguous())
        dx = torch.empty(B, Cc, h, w, dtype=dt, device=gs[0].device)
        ptrs = (C.c_void_p * len(scales))(*[g.data_ptr() for g in gs])
        _lib.check(_lib.lib().sd_ppm_pool_bwd(ptrs, _DT[dt], B * Cc, h, w, _scales_arr(scales), len(scales), dx.data_ptr(), _stream_ptr()),
                   'sd_ppm_pool_bwd')
        return dx, None


def ppm_pool(x, scales):
    """-> tuple of F.adaptive_avg_pool2d(x, s) for s in scales (contiguous fp32 / bf16 CUDA maps: see supported())."""
    return _PPMPool.apply(x, tuple(int(s) for s in scales))

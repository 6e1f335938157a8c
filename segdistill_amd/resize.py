"""Bilinear resize of NCHW maps on the HIP kernels of csrc/resize.hip, with autograd (reference mmseg/ops/wrappers.py:6-28 -> F.interpolate).

ATen's NCHW bilinear kernels run at ~3 % of HBM speed on MI355X (1.3 ms for a 268 MB map, 1.5 ms for its atomic backward: 27 of the 99 ms
of a config-4 step, 3.8 of the 26 ms of config 1); `layers.resize` routes contiguous fp32 / bf16 CUDA maps here.  Channels-last maps keep
ATen's NHWC kernel (a different, vectorised code path)."""
from __future__ import annotations

import os

import torch

from . import _lib
from .ops import _DT, _stream_ptr

ENABLED = os.environ.get('SEGDISTILL_HIP_RESIZE', '1') == '1'


def supported(x, size, mode, align_corners):
    # contiguous NCHW maps; a strided view that is NOT channels-last (a slice, an expand) is made contiguous first -- ATen would do the same
    # copy and then run its 3 %-of-HBM NCHW kernel (profiles/r03_train_step_kernels_cfg4.txt: one such call, 2.47 ms); channels-last maps keep
    # ATen's NHWC kernel
    return (ENABLED and mode == 'bilinear' and size is not None and x.is_cuda and x.dim() == 4 and x.dtype in _DT and x.numel() > 0 and len(size) == 2
            and (x.is_contiguous() or not x.is_contiguous(memory_format=torch.channels_last)))


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W, align):
        x = x.contiguous()
        B, C, h, w = x.shape
        y = torch.empty(B, C, H, W, dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().sd_resize_bilinear_fwd(x.data_ptr(), y.data_ptr(), _DT[x.dtype], B * C, h, w, H, W, int(align), _stream_ptr()),
                   'sd_resize_bilinear_fwd')
        ctx.geom = (B, C, h, w, H, W, int(align))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, h, w, H, W, align = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(B, C, h, w, dtype=dy.dtype, device=dy.device)
        L = _lib.lib()
        wsb = L.sd_resize_bilinear_bwd_workspace_bytes(B * C, h, W)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dy.device)
        _lib.check(L.sd_resize_bilinear_bwd(dy.data_ptr(), dx.data_ptr(), _DT[dy.dtype], B * C, h, w, H, W, align, ws.data_ptr(), wsb, _stream_ptr()),
                   'sd_resize_bilinear_bwd')
        return dx, None, None, None


def bilinear(x, size, align_corners=False):
    return _Bilinear.apply(x, int(size[0]), int(size[1]), bool(align_corners))

"""autograd binding of the MFMA 1x1 projection kernels (csrc/align1x1.hip)."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _DT, _require_gpu, _stream_ptr


class _Align1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, weight)
        if x.dim() != 4 or weight.dim() != 2 or weight.shape[1] != x.shape[1]:
            raise ValueError(f'align1x1: x {tuple(x.shape)} vs weight {tuple(weight.shape)}')
        if x.dtype not in _DT or weight.dtype != torch.float32:
            raise TypeError('align1x1: activations fp32/bf16, weight fp32')
        x = x.contiguous()
        w = weight.contiguous()
        b = None if bias is None else bias.contiguous().float()
        B, Cs, h, wd = x.shape
        Ct = w.shape[0]
        y = torch.empty(B, Ct, h, wd, dtype=x.dtype, device=x.device)
        rc = _lib.lib().sd_align1x1_fwd(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), _DT[x.dtype], B, Cs,
                                        Ct, h, wd, _stream_ptr())
        _lib.check(rc, 'sd_align1x1_fwd')
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cs, h, wd = x.shape
        Ct = w.shape[0]
        L = _lib.lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.sd_align1x1_bwd_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), _DT[x.dtype], B, Cs, Ct, h, wd, _stream_ptr()),
                       'sd_align1x1_bwd_data')
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(Ct, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            wsb = L.sd_align1x1_workspace_bytes(B, Cs, Ct, h, wd)
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            _lib.check(L.sd_align1x1_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(),
                                                _DT[x.dtype], B, Cs, Ct, h, wd, ws.data_ptr(), wsb, _stream_ptr()),
                       'sd_align1x1_bwd_weight')
        return dx, dw, db


def align1x1(x, weight, bias=None):
    return _Align1x1.apply(x, weight, bias)

"""nn.LayerNorm drop-in whose GPU forward/backward run the token-major HIP kernels (csrc/layernorm.hip).
Same parameters and state-dict keys (weight, bias); on CPU tensors, under autocast or for unsupported C it is
exactly nn.LayerNorm."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .ops import _DT, _stream_ptr


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        xc = x.contiguous()
        C = xc.shape[-1]
        rows = xc.numel() // C
        y = torch.empty_like(xc)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        rc = _lib.lib().sd_layernorm_fwd(xc.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _DT[xc.dtype],
                                         rows, C, float(eps), _stream_ptr())
        _lib.check(rc, 'sd_layernorm_fwd')
        ctx.save_for_backward(xc, w, mean, rstd)
        ctx.pdtype = weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        L = _lib.lib()
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        wsb = L.sd_layernorm_workspace_bytes(rows, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        rc = L.sd_layernorm_bwd(x.data_ptr(), dy.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                db.data_ptr(), _DT[x.dtype], rows, C, ws.data_ptr(), wsb, _stream_ptr())
        _lib.check(rc, 'sd_layernorm_bwd')
        return dx, dg.to(ctx.pdtype), db.to(ctx.pdtype), None


class HipLayerNorm(nn.LayerNorm):
    def forward(self, x):
        if (x.is_cuda and x.dtype in _DT and self.elementwise_affine and self.bias is not None and len(self.normalized_shape) == 1
                and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024 and x.numel() > 0):
            return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)

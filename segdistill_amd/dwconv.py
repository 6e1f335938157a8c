"""autograd binding of the token-major depth-wise 3x3 kernels (csrc/dwconv.hip)."""
from __future__ import annotations

import torch

from . import _lib, deferred
from .layers import frozen_derived
from .ops import _DT, _stream_ptr


class _DWConv3x3Tokens(torch.autograd.Function):
    """tokens [B, H*W, C] -> [B, H*W, C]; weight is nn.Conv2d(C, C, 3, 1, 1, groups=C).weight ([C,1,3,3]), bias [C]."""

    @staticmethod
    def forward(ctx, tokens, weight, bias, H, W, gelu=False):
        x = tokens.contiguous()
        B, N, C = x.shape
        assert N == H * W
        w_t = weight.detach().reshape(C, 9).float().contiguous()  # nn.Conv2d's own [C][9] layout: a view for an fp32 parameter
        b = None if bias is None else bias.detach().contiguous().float()
        y = torch.empty_like(x)
        if gelu:   # Mix-FFN: the exact-erf GELU that follows the conv in the same pass; the pre-activation is kept for its backward
            pre = torch.empty_like(x)
            rc = _lib.lib().sd_dwconv3x3_gelu_fwd_train(x.data_ptr(), w_t.data_ptr(), None if b is None else b.data_ptr(), pre.data_ptr(),
                                                        y.data_ptr(), _DT[x.dtype], B, H, W, C, _stream_ptr())
            _lib.check(rc, 'sd_dwconv3x3_gelu_fwd_train')
            ctx.save_for_backward(x, w_t, pre)
        else:
            rc = _lib.lib().sd_dwconv3x3_fwd(x.data_ptr(), w_t.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), _DT[x.dtype], B, H,
                                             W, C, _stream_ptr())
            _lib.check(rc, 'sd_dwconv3x3_fwd')
            ctx.save_for_backward(x, w_t)
        ctx.geom = (H, W, bias is not None, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_t = ctx.saved_tensors[:2]
        H, W, has_bias, wdtype = ctx.geom
        B, N, C = x.shape
        dy = dy.contiguous()
        if len(ctx.saved_tensors) == 3:
            dy = torch.ops.aten.gelu_backward(dy, ctx.saved_tensors[2], approximate='none')
        L = _lib.lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.sd_dwconv3x3_bwd_data(dy.data_ptr(), w_t.data_ptr(), dx.data_ptr(), _DT[x.dtype], B, H, W, C, _stream_ptr()),
                       'sd_dwconv3x3_bwd_data')
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            # one buffer [9*C + C]: the weight gradient in the parameter's own [C,1,3,3] layout, then the bias gradient
            buf = torch.empty(10 * C, dtype=torch.float32, device=x.device)
            wsb = L.sd_dwconv3x3_workspace_bytes(_DT[x.dtype], B, H, W, C)
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            later = deferred.enabled() and wdtype == torch.float32     # fp32 leaf parameters: combine at the end of the backward
            def launch():
                _lib.check(L.sd_dwconv3x3_bwd_weight(x.data_ptr(), dy.data_ptr(), None if later else buf.data_ptr(),
                                                     None if later else buf[9 * C:].data_ptr(), _DT[x.dtype], B, H, W, C, ws.data_ptr(), wsb,
                                                     _stream_ptr()), 'sd_dwconv3x3_bwd_weight')
            if later and deferred._WGRAD_GROUPED:
                # round 5: not even the partials are launched now -- all depth-wise filter gradients of the backward run as ONE launch at its end
                deferred.add_dw_wgrad(x, dy, ws, B, H, W, C)
                deferred.add(ws, buf, 10 * C, L.sd_dwconv3x3_wgrad_slabs(_DT[x.dtype], B, H, W, C))
            elif later:   # partials only, off the critical chain: side stream, combined when the scope ends
                deferred.side_launch(launch, x, dy, ws)
                deferred.add(ws, buf, 10 * C, L.sd_dwconv3x3_wgrad_slabs(_DT[x.dtype], B, H, W, C))
            else:
                launch()
            dw = buf[:9 * C].view(C, 1, 3, 3).to(wdtype)
            db = buf[9 * C:].to(wdtype) if has_bias else None
        return dx, dw, db, None, None, None


def supported(tokens, weight):
    if not tokens.is_cuda or tokens.dtype not in _DT or tokens.dim() != 3:
        return False
    C = tokens.shape[-1]
    if weight.shape != (C, 1, 3, 3):
        return False
    return C % (4 if tokens.dtype == torch.float32 else 8) == 0


def dwconv3x3_tokens(tokens, weight, bias, H, W):
    return _DWConv3x3Tokens.apply(tokens, weight, bias, H, W)


def dwconv3x3_gelu_tokens(tokens, weight, bias, H, W):
    """GELU(dwconv(tokens) + bias), exact erf form, with autograd: the training counterpart of the inference kernel below."""
    return _DWConv3x3Tokens.apply(tokens, weight, bias, H, W, True)


def dwconv3x3_gelu_tokens_inference(tokens, weight, bias, H, W):
    """GELU(dwconv(tokens) + bias) in one kernel; no autograd (frozen-teacher path)."""
    x = tokens.contiguous()
    B, N, C = x.shape
    w_t = (weight.detach().reshape(C, 9) if weight.dtype == torch.float32 and weight.is_contiguous()
           else frozen_derived(weight, 'dw_taps', lambda: weight.detach().reshape(C, 9).float().contiguous()))
    b = None if bias is None else bias.detach().contiguous().float()
    y = torch.empty_like(x)
    rc = _lib.lib().sd_dwconv3x3_gelu_fwd(x.data_ptr(), w_t.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), _DT[x.dtype], B, H, W,
                                          C, _stream_ptr())
    _lib.check(rc, 'sd_dwconv3x3_gelu_fwd')
    return y

"""torch.autograd bindings over the C ABI (device pointers + the current HIP stream)."""
from __future__ import annotations

import torch

from . import _lib

_DT = {torch.float32: _lib.SD_F32, torch.bfloat16: _lib.SD_BF16}


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError('segdistill_amd ops run on the GPU only (tensor is on %s); there is no CPU path' % t.device)


def _ptr(t):
    return None if t is None else t.data_ptr()


class _CGDKLFunction(torch.autograd.Function):
    """loss = alpha/rows * sum_rows KL(softmax(T_r/tau) || softmax(S_r/tau)); S,T at softmax resolution."""

    @staticmethod
    def forward(ctx, S, T, group_size, tau, alpha, perm):
        _require_gpu(S, T)
        if S.shape != T.shape or S.dim() != 4:
            raise ValueError(f'expected equal 4-D shapes, got {tuple(S.shape)} and {tuple(T.shape)}')
        if S.dtype != T.dtype or S.dtype not in _DT:
            raise TypeError(f'unsupported dtypes {S.dtype}/{T.dtype}')
        S, T = S.contiguous(), T.contiguous()
        B, Cc, H, W = S.shape
        g = int(group_size)
        G = -(-Cc // g)
        rows = B * G
        L = _lib.lib()
        if perm is not None:
            perm = perm.to(device=S.device, dtype=torch.int32).contiguous()
            if perm.numel() != Cc:
                raise ValueError('perm must have C entries')
        ws_bytes = L.sd_cgd_kl_workspace_bytes(B, Cc, H, W, g)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=S.device)
        row_lse2 = torch.empty(rows, 2, dtype=torch.float32, device=S.device)
        row_kl = torch.empty(rows, dtype=torch.float32, device=S.device)
        loss = torch.empty((), dtype=torch.float32, device=S.device)
        rc = L.sd_cgd_kl_fwd(S.data_ptr(), T.data_ptr(), _DT[S.dtype], B, Cc, H, W, g, 1.0 / float(tau), float(alpha) / rows,
                             _ptr(perm), row_lse2.data_ptr(), row_kl.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws_bytes,
                             _stream_ptr())
        _lib.check(rc, 'sd_cgd_kl_fwd')
        ctx.save_for_backward(S, T, row_lse2, perm if perm is not None else torch.empty(0, device=S.device))
        ctx.meta = (g, float(tau), float(alpha), rows, perm is not None)
        ctx.row_kl = row_kl
        ctx.mark_non_differentiable(row_kl)
        return loss, row_kl

    @staticmethod
    def backward(ctx, grad_loss, _grad_rows):
        S, T, row_lse2, perm = ctx.saved_tensors
        g, tau, alpha, rows, has_perm = ctx.meta
        B, Cc, H, W = S.shape
        dS = torch.empty_like(S)
        up = grad_loss.to(torch.float32).contiguous()
        rc = _lib.lib().sd_cgd_kl_bwd(S.data_ptr(), T.data_ptr(), _DT[S.dtype], B, Cc, H, W, g, 1.0 / tau, alpha / (rows * tau),
                                      perm.data_ptr() if has_perm else None, row_lse2.data_ptr(), up.data_ptr(), dS.data_ptr(),
                                      _stream_ptr())
        _lib.check(rc, 'sd_cgd_kl_bwd')
        return dS, None, None, None, None, None


def cgd_kl(S, T, *, group_size, tau, alpha, perm=None, return_rows=False):
    """Channel-group KL on operands already at softmax resolution (R1).  HIP only."""
    loss, rows = _CGDKLFunction.apply(S, T, group_size, tau, alpha, perm)
    return (loss, rows) if return_rows else loss

"""Bindings of the token-major Linear GEMMs (csrc/token_gemm.hip): exact-f32 MFMA forward (+ bias / GELU / residual) and input gradient."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _stream_ptr


def supported(x, weight):
    """fp32 activations and weight on the GPU, dense row-major tokens."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2 and x.shape[-1] == weight.shape[1]
            and x.numel() > 0)


def _weight(weight):
    """-> (tensor to pass, row stride): a row-strided view with unit column stride (a column block of a wider matrix) goes in as it is."""
    if weight.stride(1) == 1 and weight.stride(0) >= weight.shape[1]:
        return weight, weight.stride(0)
    w = weight.contiguous()
    return w, w.shape[1]


def _rows(x):
    x2 = x.reshape(-1, x.shape[-1])
    return x2 if x2.is_contiguous() else x2.contiguous()


def linear_fwd(x, weight, bias=None, residual=None, act=None, split_bf16=False):
    """act(x . W^T + bias) (+ residual); x [..., K] fp32, W [N, K] fp32 -> [..., N] fp32.  split_bf16: the bf16x3 arithmetic of
    csrc/token_gemm.hip (fp32-level accuracy on the bf16 matrix pipe)."""
    x2 = _rows(x)
    w, ldw = _weight(weight)
    T, K = x2.shape
    N = w.shape[0]
    y = torch.empty(T, N, dtype=torch.float32, device=x.device)
    r2 = None
    if residual is not None:
        r2 = _rows(residual)
        if r2.shape != y.shape or r2.dtype != torch.float32:
            raise ValueError('residual must be fp32 of the output shape')
    b = None if bias is None else (bias if bias.dtype == torch.float32 and bias.is_contiguous() else bias.float().contiguous())
    rc = _lib.lib().sd_linear_fwd(x2.data_ptr(), w.data_ptr(), ldw, None if b is None else b.data_ptr(), None if r2 is None else r2.data_ptr(),
                                  y.data_ptr(), _lib.SD_F32, T, K, N, 1 if act == 'gelu' else 0, 1 if split_bf16 else 0, _stream_ptr())
    _lib.check(rc, 'sd_linear_fwd')
    return y.reshape(*x.shape[:-1], N)


def linear_bwd_data(dy, weight, split_bf16=False):
    """dy . W; dy [..., N] fp32, W [N, K] fp32 -> [..., K] fp32."""
    d2 = _rows(dy)
    w, ldw = _weight(weight)
    T, N = d2.shape
    K = w.shape[1]
    dx = torch.empty(T, K, dtype=torch.float32, device=dy.device)
    rc = _lib.lib().sd_linear_bwd_data(d2.data_ptr(), w.data_ptr(), ldw, dx.data_ptr(), _lib.SD_F32, T, K, N, 1 if split_bf16 else 0, _stream_ptr())
    _lib.check(rc, 'sd_linear_bwd_data')
    return dx.reshape(*dy.shape[:-1], K)


def linear_fwd_planes(x, weight, planes, bias=None, residual=None):
    """x . W^T + bias (+ residual) in split-bf16 arithmetic with the weight's pre-split forward planes (segdistill_amd/planes.py)."""
    x2 = _rows(x)
    T, K = x2.shape
    N = weight.shape[0]
    y = torch.empty(T, N, dtype=torch.float32, device=x.device)
    r2 = None
    if residual is not None:
        r2 = _rows(residual)
        if r2.shape != y.shape or r2.dtype != torch.float32:
            raise ValueError('residual must be fp32 of the output shape')
    b = None if bias is None else (bias if bias.dtype == torch.float32 and bias.is_contiguous() else bias.float().contiguous())
    rc = _lib.lib().sd_linear_fwd_planes(x2.data_ptr(), planes.data_ptr(), None if b is None else b.data_ptr(), None if r2 is None else r2.data_ptr(),
                                         y.data_ptr(), _lib.SD_F32, T, K, N, _stream_ptr())
    _lib.check(rc, 'sd_linear_fwd_planes')
    return y.reshape(*x.shape[:-1], N)


def linear_bwd_data_planes(dy, weight, planes):
    """dy . W with the weight's pre-split bwd planes."""
    d2 = _rows(dy)
    T, N = d2.shape
    K = weight.shape[1]
    dx = torch.empty(T, K, dtype=torch.float32, device=dy.device)
    rc = _lib.lib().sd_linear_bwd_data_planes(d2.data_ptr(), planes.data_ptr(), dx.data_ptr(), _lib.SD_F32, T, K, N, _stream_ptr())
    _lib.check(rc, 'sd_linear_bwd_data_planes')
    return dx.reshape(*dy.shape[:-1], K)

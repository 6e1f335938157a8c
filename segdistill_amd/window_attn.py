"""Window attention of a Swin block without an autograd graph (the frozen teacher of BASELINE config 4) as ONE kernel over the qkv Linear's
output (csrc/window_attn.hip; reference mmseg/models/backbones/swin_transformer.py:119-153): relative position bias and shift mask are read
from their small tables inside the kernel, q / k / v are not permuted and the output comes out token-major for the projection."""
from __future__ import annotations

import os

import torch

from . import _lib
from .ops import _DT, _stream_ptr

ENABLED = os.environ.get('SEGDISTILL_WINDOW_ATTN', '1') == '1'       # A/B: 0 = F.scaled_dot_product_attention with a materialised additive tensor


def supported(qkv, n, heads, head_dim):
    """qkv: the qkv Linear's output [windows, N, 3*heads*D]."""
    return bool(ENABLED and qkv.is_cuda and qkv.dtype == torch.float32 and qkv.numel() > 0 and qkv.dim() == 3
                and _lib.lib().sd_window_attn_supported(int(n), int(head_dim)))


def forward(qkv, bias_t, mask_t, heads, scale):
    """qkv [windows, N, 3*heads*D] (contiguous), bias_t [heads, N, N], mask_t [nW, N, N] or None (both transposed, fp32) -> [windows, N, heads*D]."""
    bw, n, c3 = qkv.shape
    c = c3 // 3
    assert qkv.is_contiguous() and bias_t.is_contiguous() and bias_t.dtype == torch.float32 and bias_t.shape == (heads, n, n)
    nw = 0
    if mask_t is not None:
        assert mask_t.is_contiguous() and mask_t.dtype == torch.float32 and mask_t.shape[1:] == (n, n)
        nw = mask_t.shape[0]
    out = torch.empty(bw, n, c, dtype=qkv.dtype, device=qkv.device)
    rc = _lib.lib().sd_window_attn_fwd(qkv.data_ptr(), bias_t.data_ptr(), None if mask_t is None else mask_t.data_ptr(), out.data_ptr(),
                                       _DT[qkv.dtype], bw, nw, heads, n, c // heads, float(scale), _stream_ptr())
    _lib.check(rc, 'sd_window_attn_fwd')
    return out

"""Window attention of a Swin block without an autograd graph (the frozen teacher of BASELINE config 4) as ONE kernel over the qkv Linear's
output (csrc/window_attn.hip; reference mmseg/models/backbones/swin_transformer.py:119-153): relative position bias and shift mask are read
from small packed tables inside the kernel, q / k / v are not permuted and the output comes out token-major for the projection."""
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


def pack_tables(tables, pad_key=0.0, want_flags=False):
    """[n, 49, 49] additive tables in the reference's orientation ([query, key]) -> ([n, 4096] in the MFMA kernel's accumulator layout,
    int32 flags [n] = table has a nonzero entry, or None)."""
    t = tables.detach().float().contiguous()
    n = t.shape[0]
    assert t.shape[1:] == (49, 49)
    L = _lib.lib()
    packed = torch.empty(n, int(L.sd_window_attn_packed_floats()), dtype=torch.float32, device=t.device)
    flags = torch.empty(n, dtype=torch.int32, device=t.device) if want_flags else None
    _lib.check(L.sd_window_attn_pack(t.data_ptr(), packed.data_ptr(), None if flags is None else flags.data_ptr(), n, 49, float(pad_key), _stream_ptr()),
               'sd_window_attn_pack')
    return packed, flags


def forward_packed(qkv, bias_p, mask_p, mask_flags, heads, scale):
    """qkv [windows, 49, 3*heads*32] (contiguous fp32); bias_p = pack_tables(bias [heads, 49, 49], -inf)[0]; mask_p, mask_flags =
    pack_tables(mask [nW, 49, 49], 0, True) or (None, None) -> [windows, 49, heads*32]."""
    bw, n, c3 = qkv.shape
    c = c3 // 3
    assert qkv.is_contiguous() and bias_p.shape[0] == heads
    out = torch.empty(bw, n, c, dtype=qkv.dtype, device=qkv.device)
    rc = _lib.lib().sd_window_attn_fwd_packed(qkv.data_ptr(), bias_p.data_ptr(), None if mask_p is None else mask_p.data_ptr(),
                                              None if mask_p is None else mask_flags.data_ptr(), out.data_ptr(), _DT[qkv.dtype], bw,
                                              0 if mask_p is None else mask_p.shape[0], heads, n, c // heads, float(scale), _stream_ptr())
    _lib.check(rc, 'sd_window_attn_fwd_packed')
    return out

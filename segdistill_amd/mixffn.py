"""The second half of a frozen Mix-FFN as one kernel (csrc/mixffn_tail.hip): fc2(GELU(dwconv3x3(h) + b)) on token-major fp32 activations.

reference mix_transformer.py:20-55 (`x = self.dwconv(x, H, W); x = self.act(x); x = self.drop(x); x = self.fc2(x)`), eval mode, no autograd
(the teacher's stages 1-2): the activated hidden map is never written.  Same convolution / GELU arithmetic as dwconv.py's inference kernel; fp32
storage: the product in split-bf16 arithmetic as linear.py's fp32 kernels; bf16 activations under autocast: a bf16 product on the weight's bf16
rounding with fp32 accumulation, as the bf16 GEMM of the two-kernel route."""
from __future__ import annotations

import os

import torch

from . import _lib
from .layers import frozen_derived
from .ops import _DT, _stream_ptr

_ENABLED = os.environ.get('SEGDISTILL_MIXFFN_TAIL', '1') == '1'      # A/B: 0 = the depthwise kernel, then the GEMM
_MIN_TOKENS = int(os.environ.get('SEGDISTILL_MIXFFN_TAIL_MIN_TOKENS', '16384'))   # below: too few 128-token patches to fill the CUs


def usable(h, conv, fc2, hw):
    """h: fc1's fp32 output [B, H*W, hidden] on the GPU with autograd off; conv the depthwise 3x3 (+ bias), fc2 the output Linear (+ bias)."""
    if not (_ENABLED and h.is_cuda and h.dim() == 3 and not torch.is_grad_enabled()):
        return False
    # fp32 storage without autocast, or bf16 activations under bf16 autocast (fc2 would then run as a bf16 product on the weight's bf16 rounding)
    amp = torch.is_autocast_enabled()
    if h.dtype == torch.float32 and os.environ.get('SEGDISTILL_SPLIT_BF16', '1') != '1':
        return False      # exact-f32 mode (bench.py's value_exact_f32): this kernel's fp32 product is split-bf16 arithmetic
    if not ((h.dtype == torch.float32 and not amp) or (h.dtype == torch.bfloat16 and amp and torch.get_autocast_dtype('cuda') == torch.bfloat16)):
        return False
    if conv.bias is None or fc2.bias is None or conv.weight.dtype != torch.float32 or fc2.weight.dtype != torch.float32:
        return False
    if fc2._forward_hooks or fc2._forward_pre_hooks:
        return False
    B, N, C = h.shape
    H, W = int(hw[0]), int(hw[1])
    return (N == H * W and B * N >= _MIN_TOKENS and tuple(conv.weight.shape) == (C, 1, 3, 3) and fc2.in_features == C
            and bool(_lib.lib().sd_mixffn_tail_supported(H, W, C, fc2.out_features)))


def tail(h, conv, fc2, hw):
    h = h.contiguous()
    B, N, C = h.shape
    w_t = (conv.weight.detach().reshape(C, 9) if conv.weight.is_contiguous()
           else frozen_derived(conv.weight, 'dw_taps', lambda: conv.weight.detach().reshape(C, 9).float().contiguous()))
    w2 = fc2.weight.detach()
    w2 = w2 if w2.is_contiguous() else w2.contiguous()
    y = torch.empty(B, N, fc2.out_features, dtype=h.dtype, device=h.device)
    rc = _lib.lib().sd_mixffn_tail(h.data_ptr(), w_t.data_ptr(), conv.bias.detach().data_ptr(), w2.data_ptr(), fc2.bias.detach().data_ptr(),
                                   y.data_ptr(), _DT[h.dtype], B, int(hw[0]), int(hw[1]), C, fc2.out_features, _stream_ptr())
    _lib.check(rc, 'sd_mixffn_tail')
    return y

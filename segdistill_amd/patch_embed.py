"""The MiT patch-embedding convolutions as window gather + token GEMM (csrc/patch_embed.hip; reference mix_transformer.py:185-215).

MIOpen's filter-gradient kernels for these shapes accumulate with float atomics (two runs of one step differ in the last bits) and under the
reference's launch mode `--deterministic` (tools/dist_train.sh:8) it falls back to naive kernels: 112 ms per config-2 step instead of 9.7.  Here
the k x k windows are gathered once into a token-major matrix and the three products run on the token GEMMs of the encoders (the filter
gradient joins the backward's grouped launch), so the step is run-to-run bit-identical WITHOUT a switch and the projection's output already is
the token map."""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import _lib
from .linear import token_linear
from .ops import _DT, _stream_ptr

# A/B: 0 = the convolution through MIOpen (rounds 1-5)
_ENABLED = os.environ.get('SEGDISTILL_PATCH_EMBED_GEMM', '1') == '1'


def _geometry(x, conv):
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    B, Cin, H, W = x.shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    K = k * k * Cin
    return k, s, p, B, Cin, H, W, Ho, Wo, K, -(-K // 8) * 8


def supported(x, conv):
    """x: logical [B, Cin, H, W] on the GPU (a channels-last view of tokens, or the NCHW image), fp32 / bf16 storage; a plain square convolution."""
    return (_ENABLED and x.is_cuda and x.dim() == 4 and x.dtype in _DT and conv.weight.dtype == torch.float32
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1] and isinstance(conv.padding, tuple)
            and conv.padding[0] == conv.padding[1] and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and not (conv._forward_hooks or conv._forward_pre_hooks)
            and (not torch.is_autocast_enabled() or torch.get_autocast_dtype('cuda') == torch.bfloat16) and x.numel() > 0)


class _Im2Col(torch.autograd.Function):
    """col [B, Ho * Wo, Kp] of x's k x k windows; backward = the transposed gather (deterministic, no atomics), returned in channels-last storage."""

    @staticmethod
    def forward(ctx, x, k, s, p, Ho, Wo, Kp):
        B, Cin, H, W = x.shape
        col = torch.empty(B, Ho * Wo, Kp, dtype=x.dtype, device=x.device)
        sb, sc, sy, sx = x.stride()
        _lib.check(_lib.lib().sd_im2col_tokens(x.data_ptr(), col.data_ptr(), _DT[x.dtype], B, H, W, Cin, sb, sc, sy, sx, k, s, p, Ho, Wo, Kp, _stream_ptr()),
                   'sd_im2col_tokens')
        ctx.geom = (B, Cin, H, W, k, s, p, Ho, Wo, Kp)
        return col

    @staticmethod
    def backward(ctx, dcol):
        B, Cin, H, W, k, s, p, Ho, Wo, Kp = ctx.geom
        dcol = dcol.contiguous()
        epc = 16 // dcol.element_size()
        if Cin % epc:
            # not a shape of the networks here (the image needs no gradient); the general form through ATen's fold
            d = dcol[..., :k * k * Cin].reshape(B, Ho * Wo, k * k, Cin).permute(0, 3, 2, 1).reshape(B, Cin * k * k, Ho * Wo)
            return F.fold(d.float(), (H, W), k, padding=p, stride=s).to(dcol.dtype), None, None, None, None, None, None
        dx = torch.empty(B, H, W, Cin, dtype=dcol.dtype, device=dcol.device)
        _lib.check(_lib.lib().sd_col2im_tokens(dcol.data_ptr(), dx.data_ptr(), _DT[dcol.dtype], B, H, W, Cin, k, s, p, Ho, Wo, Kp, _stream_ptr()),
                   'sd_col2im_tokens')
        return dx.permute(0, 3, 1, 2), None, None, None, None, None, None


def im2col_tokens(x, k, s, p):
    B, Cin, H, W = x.shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    Kp = -(-(k * k * Cin) // 8) * 8
    return _Im2Col.apply(x, k, s, p, Ho, Wo, Kp), (Ho, Wo)


def patch_embed_tokens(x, conv):
    """conv(x) as tokens: ([B, Ho * Wo, out], (Ho, Wo)).  The filter is read through its (ky, kx, ci)-ordered matrix view -- a VIEW of the parameter
    when it is kept in channels-last storage (backbones/mit.py does), so its gradient goes straight from the backward's grouped launch to the leaf."""
    from .layers import frozen_derived
    k, s, p, B, Cin, H, W, Ho, Wo, K, Kp = _geometry(x, conv)
    w = conv.weight
    out = w.shape[0]
    if torch.is_autocast_enabled() and x.dtype == torch.float32:
        x = x.to(torch.get_autocast_dtype('cuda'))            # what autocast's conv2d does with the image; the windows are gathered in bf16
    col = _Im2Col.apply(x, k, s, p, Ho, Wo, Kp)
    w2 = w.permute(0, 2, 3, 1).reshape(out, K)
    is_view = w2.data_ptr() == w.data_ptr() and w2._base is not None
    if Kp != K:
        # stage 1 (3 x 7 x 7 = 147 -> 152 columns): a zero-padded copy of the 32 x 147 matrix; its gradient is sliced back by autograd at once,
        # so it cannot wait for the deferred launch
        w2 = frozen_derived(w, ('patch_embed_pad', Kp), lambda: F.pad(w.permute(0, 2, 3, 1).reshape(out, K), (0, Kp - K)))
        is_view = False
    elif not is_view:
        w2 = frozen_derived(w, 'patch_embed_matrix', lambda: w.permute(0, 2, 3, 1).reshape(out, K).contiguous())
    return token_linear(col, w2, conv.bias, defer_ok=is_view, defer_bias_ok=True), (Ho, Wo)

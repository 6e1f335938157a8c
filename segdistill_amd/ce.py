"""autograd binding of the fused up-sample + cross-entropy kernels (csrc/ce_up.hip)."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _DT, _stream_ptr


def supported(logits, label_hw):
    if not logits.is_cuda or logits.dtype not in _DT or logits.dim() != 4:
        return False
    h, w = logits.shape[2:]
    return bool(_lib.lib().sd_ce_up_supported(int(h), int(w), int(label_hw[0]), int(label_hw[1])))


class _FusedCEUp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, ignore_index):
        x = logits.contiguous()
        B, C, h, w = x.shape
        H, W = label.shape[-2:]
        lab = label.reshape(B, H, W).to(torch.int32).contiguous()
        loss_pix = torch.empty(B, H, W, dtype=torch.float32, device=x.device)
        lse2 = torch.empty(B, H, W, dtype=torch.float32, device=x.device)
        correct = torch.empty(1, dtype=torch.int32, device=x.device)
        rc = _lib.lib().sd_ce_up_fwd(x.data_ptr(), lab.data_ptr(), loss_pix.data_ptr(), lse2.data_ptr(), correct.data_ptr(), _DT[x.dtype], B, C,
                                     h, w, H, W, int(ignore_index), _stream_ptr())
        _lib.check(rc, 'sd_ce_up_fwd')
        ctx.save_for_backward(x, lab, lse2)
        ctx.ignore_index = int(ignore_index)
        ctx.mark_non_differentiable(correct)
        return loss_pix, correct

    @staticmethod
    def backward(ctx, g, _gc):
        x, lab, lse2 = ctx.saved_tensors
        B, C, h, w = x.shape
        H, W = lab.shape[-2:]
        dx = torch.empty_like(x)
        uniform = all(s == 0 for s in g.stride())  # e.g. the gradient of .mean(): one value broadcast over the map
        up = (g.reshape(-1)[:1] if uniform else g).to(torch.float32).contiguous()
        rc = _lib.lib().sd_ce_up_bwd(x.data_ptr(), lab.data_ptr(), lse2.data_ptr(), up.data_ptr(), 0 if uniform else 1, 1.0, dx.data_ptr(),
                                     _DT[x.dtype], B, C, h, w, H, W, ctx.ignore_index, _stream_ptr())
        _lib.check(rc, 'sd_ce_up_bwd')
        return dx, None, None


def fused_ce_up(logits, label, ignore_index=255):
    """-> (per-pixel CE [B,H,W] at label resolution, 0 on ignored pixels; int32[1] count of top-1 hits)."""
    return _FusedCEUp.apply(logits, label, ignore_index)

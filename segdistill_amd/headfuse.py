"""autograd binding of the SegFormer-head up-sample-and-sum kernels (csrc/headfuse.hip)."""
from __future__ import annotations

import os

import torch

from . import _lib
from .ops import _DT, _stream_ptr


def supported(zs, sizes):
    """zs: four token-major tensors [B, h_i*w_i, E] (finest first); sizes: their (h_i, w_i)."""
    z1 = zs[0]
    if not z1.is_cuda or z1.dtype not in _DT or any(z.dtype != z1.dtype or z.dim() != 3 for z in zs):
        return False
    E = z1.shape[-1]
    if E % (4 if z1.dtype == torch.float32 else 8):
        return False
    H, W = sizes[0]
    for (h, w) in sizes[1:]:
        if h <= 0 or w <= 0 or H % h or W % w or H // h != W // w or H // h not in (2, 4, 8):
            return False
    return True


class _UpSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2, z3, z4, bias, sizes):
        zs = [z.contiguous() for z in (z1, z2, z3, z4)]
        B, _, E = zs[0].shape
        (H, W) = sizes[0]
        fs = [H // h for (h, w) in sizes[1:]]
        y = torch.empty_like(zs[0])
        b = None if bias is None else bias.detach().float().contiguous()
        rc = _lib.lib().sd_upsum_fwd(zs[0].data_ptr(), zs[1].data_ptr(), zs[2].data_ptr(), zs[3].data_ptr(), None if b is None else b.data_ptr(),
                                     y.data_ptr(), _DT[y.dtype], B, H, W, E, fs[0], fs[1], fs[2], _stream_ptr())
        _lib.check(rc, 'sd_upsum_fwd')
        ctx.sizes, ctx.fs, ctx.has_bias = sizes, fs, bias is not None
        ctx.bias_dtype = None if bias is None else bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, _, E = dy.shape
        L = _lib.lib()
        grads = [dy if ctx.needs_input_grad[0] else None]
        H, W = ctx.sizes[0]
        if (tuple(ctx.fs) == (2, 4, 8) and all(ctx.needs_input_grad[1:4]) and H % 8 == 0 and W % 8 == 0 and E % 64 == 0 and W <= 512):
            # all three branches from one read of dy (two separable passes, csrc/headfuse.hip: sd_upsum_bwd3)
            dzs = [torch.empty(B, h * w, E, dtype=dy.dtype, device=dy.device) for (h, w) in ctx.sizes[1:]]
            wsb = L.sd_upsum_bwd3_workspace_bytes(B, H, W, E)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dy.device)
            _lib.check(L.sd_upsum_bwd3(dy.data_ptr(), dzs[0].data_ptr(), dzs[1].data_ptr(), dzs[2].data_ptr(), _DT[dy.dtype], B, H, W, E, ws.data_ptr(), wsb,
                                       _stream_ptr()), 'sd_upsum_bwd3')
            db = dy.sum(dim=(0, 1)).to(ctx.bias_dtype) if (ctx.has_bias and ctx.needs_input_grad[4]) else None
            return grads[0], dzs[0], dzs[1], dzs[2], db, None
        for i, ((h, w), F) in enumerate(zip(ctx.sizes[1:], ctx.fs)):
            if not ctx.needs_input_grad[i + 1]:
                grads.append(None)
                continue
            dz = torch.empty(B, h * w, E, dtype=dy.dtype, device=dy.device)
            _lib.check(L.sd_upsum_bwd(dy.data_ptr(), dz.data_ptr(), _DT[dy.dtype], B, h, w, E, F, _stream_ptr()), 'sd_upsum_bwd')
            grads.append(dz)
        db = dy.sum(dim=(0, 1)).to(ctx.bias_dtype) if (ctx.has_bias and ctx.needs_input_grad[4]) else None
        return grads[0], grads[1], grads[2], grads[3], db, None


def upsum(z1, z2, z3, z4, bias, sizes):
    """y[B, H*W, E] = z1 + up(z2) + up(z3) + up(z4) + bias, all token-major."""
    return _UpSum.apply(z1, z2, z3, z4, bias, tuple(tuple(int(v) for v in s) for s in sizes))


def upsum_affine_inference(zs, bias, sizes, scale, shift, relu=True):
    """No-grad form for a frozen network: y = relu((z1 + up(z2) + up(z3) + up(z4) + bias) * scale + shift) in one pass -- the sum,
    the eval-mode BatchNorm folded to (scale, shift) and the ReLU.  zs: four token-major tensors, finest first."""
    zs = [z.contiguous() for z in zs]
    B, _, E = zs[0].shape
    (H, W) = sizes[0]
    fs = [H // h for (h, w) in sizes[1:]]
    y = torch.empty_like(zs[0])
    b = None if bias is None else bias.detach().float().contiguous()
    sc, sh = scale.detach().float().contiguous(), shift.detach().float().contiguous()
    rc = _lib.lib().sd_upsum_affine_fwd(zs[0].data_ptr(), zs[1].data_ptr(), zs[2].data_ptr(), zs[3].data_ptr(), None if b is None else b.data_ptr(),
                                        sc.data_ptr(), sh.data_ptr(), 1 if relu else 0, y.data_ptr(), _DT[y.dtype], B, H, W, E, fs[0], fs[1],
                                        fs[2], _stream_ptr())
    _lib.check(rc, 'sd_upsum_affine_fwd')
    return y


_HEAD_TAIL = os.environ.get('SEGDISTILL_HEAD_TAIL', '1') == '1'      # A/B: 0 = upsum_affine_inference, then the linear_pred product


def head_tail_supported(zs, sizes, classes):
    """The frozen head's sum + norm + ReLU + linear_pred as ONE kernel (csrc/head_tail.hip): fp32 token-major branch maps of the SegFormer
    geometry (factors 2, 4, 8), no autograd, no autocast."""
    if not (_HEAD_TAIL and supported(zs, sizes) and zs[0].dtype == torch.float32 and not torch.is_grad_enabled() and not torch.is_autocast_enabled()):
        return False
    if os.environ.get('SEGDISTILL_SPLIT_BF16', '1') != '1':
        return False      # exact-f32 mode (bench.py's value_exact_f32): this kernel's product is split-bf16 arithmetic
    (H, W) = sizes[0]
    if [H // h for (h, w) in sizes[1:]] != [2, 4, 8]:
        return False
    return bool(_lib.lib().sd_head_tail_supported(int(H), int(W), int(zs[0].shape[-1]), int(classes)))


def head_tail(zs, sizes, fuse_bias, scale, shift, pred_weight2d, pred_bias):
    """-> logits [B, classes, H, W] = W_p . relu(scale * (z1 + up(z2) + up(z3) + up(z4) + fuse_bias) + shift) + pred_bias; the summed map is never
    written.  pred_weight2d: linear_pred's weight as [classes, E] (a view of the frozen parameter: its row planes are cached on it)."""
    from . import planes
    zs = [z.contiguous() for z in zs]
    B, _, E = zs[0].shape
    (H, W) = sizes[0]
    classes = pred_weight2d.shape[0]
    rows = planes.get(pred_weight2d, 'rows')
    fb = None if fuse_bias is None else fuse_bias.detach().float().contiguous()
    pb = None if pred_bias is None else pred_bias.detach().float().contiguous()
    sc, sh = scale.detach().float().contiguous(), shift.detach().float().contiguous()
    out = torch.empty(B, classes, H, W, dtype=torch.float32, device=zs[0].device)
    rc = _lib.lib().sd_head_tail_f32(zs[0].data_ptr(), zs[1].data_ptr(), zs[2].data_ptr(), zs[3].data_ptr(), None if fb is None else fb.data_ptr(),
                                     sc.data_ptr(), sh.data_ptr(), rows.data_ptr(), None if pb is None else pb.data_ptr(), out.data_ptr(), B, int(H),
                                     int(W), E, classes, _stream_ptr())
    _lib.check(rc, 'sd_head_tail_f32')
    return out

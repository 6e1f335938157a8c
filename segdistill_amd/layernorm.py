"""nn.LayerNorm drop-in whose GPU forward/backward run the token-major HIP kernels (csrc/layernorm.hip).
Same parameters and state-dict keys (weight, bias); on CPU tensors, under autocast or for unsupported C it is
exactly nn.LayerNorm."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, deferred
from .ops import _DT, _stream_ptr


def _defer(param_dtype):
    """Leave the dgamma / dbeta partials to the enclosing deferred scope?  Only when the fp32 results go to the parameters as they
    are (a dtype cast would read them before they exist)."""
    return deferred.enabled() and param_dtype == torch.float32


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
        dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)   # [dgamma; dbeta]
        wsb = L.sd_layernorm_workspace_bytes(rows, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        later = _defer(ctx.pdtype)
        rc = L.sd_layernorm_bwd(x.data_ptr(), dy.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                None if later else dgb[0].data_ptr(), None if later else dgb[1].data_ptr(), _DT[x.dtype], rows, C, ws.data_ptr(),
                                wsb, _stream_ptr())
        _lib.check(rc, 'sd_layernorm_bwd')
        if later:
            deferred.add(ws, dgb, 2 * C, L.sd_layernorm_bwd_blocks(rows, C))
        return dx, dgb[0].to(ctx.pdtype), dgb[1].to(ctx.pdtype), None


class HipLayerNorm(nn.LayerNorm):
    def forward(self, x):
        if (x.is_cuda and x.dtype in _DT and self.elementwise_affine and self.bias is not None and len(self.normalized_shape) == 1
                and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024 and x.numel() > 0):
            return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)


class _AddLayerNormFn(torch.autograd.Function):
    """(xsum, y) = (x + s*res, LayerNorm(x + s*res)); s = per-sample stochastic-depth factor or None."""

    @staticmethod
    def forward(ctx, x, res, scale, weight, bias, eps):
        xc, rc_ = x.contiguous(), res.contiguous()
        C = xc.shape[-1]
        rows = xc.numel() // C
        xsum, y = torch.empty_like(xc), torch.empty_like(xc)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        sc = None if scale is None else scale.detach().float().contiguous()
        rps = rows // xc.shape[0]
        rc = _lib.lib().sd_add_layernorm_fwd(xc.data_ptr(), rc_.data_ptr(), None if sc is None else sc.data_ptr(), rps, xsum.data_ptr(),
                                             w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _DT[xc.dtype], rows, C,
                                             float(eps), _stream_ptr())
        _lib.check(rc, 'sd_add_layernorm_fwd')
        ctx.save_for_backward(xsum, w, mean, rstd, sc)
        ctx.pdtype, ctx.rps = weight.dtype, rps
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(mean, rstd)
        return xsum, y

    @staticmethod
    def backward(ctx, g_xsum, g_y):
        xsum, w, mean, rstd, sc = ctx.saved_tensors
        C = xsum.shape[-1]
        rows = xsum.numel() // C
        if g_y is None:      # the normalised output was not used: plain residual gradient
            if g_xsum is None:
                return None, None, None, None, None, None
            g_res = g_xsum if sc is None else g_xsum * sc.view(-1, *([1] * (g_xsum.dim() - 1))).to(g_xsum.dtype)
            return g_xsum, g_res, None, None, None, None
        L = _lib.lib()
        dy = g_y.contiguous()
        dres = None if g_xsum is None else g_xsum.contiguous()
        dx = torch.empty_like(xsum)
        dr = None if sc is None else torch.empty_like(xsum)
        dgb = torch.empty(2, C, dtype=torch.float32, device=xsum.device)   # [dgamma; dbeta]
        wsb = L.sd_layernorm_workspace_bytes(rows, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=xsum.device)
        later = _defer(ctx.pdtype)
        rc = L.sd_add_layernorm_bwd(xsum.data_ptr(), dy.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                    None if dres is None else dres.data_ptr(), None if sc is None else sc.data_ptr(), ctx.rps, dx.data_ptr(),
                                    None if dr is None else dr.data_ptr(), None if later else dgb[0].data_ptr(),
                                    None if later else dgb[1].data_ptr(), _DT[xsum.dtype], rows, C, ws.data_ptr(), wsb, _stream_ptr())
        _lib.check(rc, 'sd_add_layernorm_bwd')
        if later:
            deferred.add(ws, dgb, 2 * C, L.sd_layernorm_bwd_blocks(rows, C))
        return dx, (dx if dr is None else dr), None, dgb[0].to(ctx.pdtype), dgb[1].to(ctx.pdtype), None


def add_layernorm_supported(x, res, norm):
    return (isinstance(norm, HipLayerNorm) and x.is_cuda and x.dtype in _DT and res.dtype == x.dtype and res.shape == x.shape and x.dim() == 3
            and norm.elementwise_affine and norm.bias is not None and len(norm.normalized_shape) == 1 and norm.normalized_shape[0] == x.shape[-1]
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024 and x.numel() > 0
            and not (norm._forward_hooks or norm._forward_pre_hooks or norm._backward_hooks))


def add_layernorm(x, res, norm, scale=None):
    """x [B, N, C] + scale[b] * res, and `norm` applied to the sum: returns (xsum, normed).  One kernel each way."""
    if res.dtype != x.dtype:   # autocast can hand over a bf16 branch output for an fp32 residual stream (or the reverse)
        dt = torch.promote_types(x.dtype, res.dtype)
        x, res = x.to(dt), res.to(dt)
    return _AddLayerNormFn.apply(x, res, scale, norm.weight, norm.bias, norm.eps)


# ---- inference LayerNorm with row maps: the Swin blocks of a frozen network (round 4) ---------------------------------------------------------------
def map_supported(x, norm):
    return (isinstance(norm, HipLayerNorm) and x.is_cuda and x.dtype in _DT and x.dim() == 3 and norm.elementwise_affine and norm.bias is not None
            and len(norm.normalized_shape) == 1 and norm.normalized_shape[0] == x.shape[-1] and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024
            and x.numel() > 0 and not (torch.is_grad_enabled() and (x.requires_grad or norm.weight.requires_grad))
            and not (norm._forward_hooks or norm._forward_pre_hooks))


def layernorm_map(x, norm, res=None, x_map=None, res_map=None, rows_out=None):
    """No-graph LayerNorm over gathered rows (csrc/layernorm.hip::ln_map_fwd).  x [B, L, C]; per image, output row r normalises
    x[x_map[r]] (+ res[res_map[r]], default the same row as x) -- x_map value L = a zero output row (window padding).  Returns
    (xsum [B, L, C] = x + gathered res, or None without res;  y [B, rows_out, C])."""
    B, L, C = x.shape
    xc = x.contiguous()
    n_out = int(rows_out if rows_out is not None else (x_map.numel() if x_map is not None else L))
    y = torch.empty(B, n_out, C, dtype=x.dtype, device=x.device)
    rc_, xsum, L_res = None, None, 0
    if res is not None:
        rc_ = (res if res.dtype == x.dtype else res.to(x.dtype)).contiguous()
        L_res = rc_.shape[1]
        xsum = torch.empty_like(xc)
    from .layers import frozen_derived
    w = norm.weight if norm.weight.dtype == torch.float32 else frozen_derived(norm.weight, 'f32', lambda: norm.weight.detach().float())
    b = norm.bias if norm.bias.dtype == torch.float32 else frozen_derived(norm.bias, 'f32', lambda: norm.bias.detach().float())
    rc = _lib.lib().sd_layernorm_map_fwd(xc.data_ptr(), None if rc_ is None else rc_.data_ptr(), None if xsum is None else xsum.data_ptr(),
                                         w.data_ptr(), b.data_ptr(), y.data_ptr(), None if x_map is None else x_map.data_ptr(),
                                         None if res_map is None else res_map.data_ptr(), _DT[xc.dtype], B, n_out, L, L_res, C, float(norm.eps),
                                         _stream_ptr())
    _lib.check(rc, 'sd_layernorm_map_fwd')
    return xsum, y


# ---- LayerNorm whose consumer is a spatial-reduction attention: the output is also produced in the SR conv's patch order (round 3) -------------
def patch_supported(x, hw, r):
    return (x.is_cuda and x.dtype in _DT and x.dim() == 3 and r > 1 and x.shape[1] == hw[0] * hw[1]
            and bool(_lib.lib().sd_layernorm_patch_supported(int(hw[0]), int(hw[1]), int(r))))


def _patch_fwd(x, res, sc, rps, w, b, eps, H, W, r):
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    yp = torch.empty(x.shape[0], (H // r) * (W // r), r * r * C, dtype=x.dtype, device=x.device)
    xsum = None if res is None else torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    rc = _lib.lib().sd_add_layernorm_patch_fwd(x.data_ptr(), None if res is None else res.data_ptr(), None if sc is None else sc.data_ptr(), rps,
                                               None if xsum is None else xsum.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), yp.data_ptr(),
                                               mean.data_ptr(), rstd.data_ptr(), _DT[x.dtype], rows, C, float(eps), H, W, r, _stream_ptr())
    _lib.check(rc, 'sd_add_layernorm_patch_fwd')
    return xsum, y, yp, mean, rstd


def _patch_bwd(ctx, xn, w, mean, rstd, sc, g_xsum, g_y, g_p, with_res):
    """shared backward: dx (and dr) of the normalised row from the token-order gradient g_y plus the patch-order gradient g_p."""
    C = xn.shape[-1]
    rows = xn.numel() // C
    L = _lib.lib()
    dy = torch.zeros_like(xn) if g_y is None else g_y.contiguous()
    dp = None if g_p is None else g_p.contiguous()
    dres = None if g_xsum is None else g_xsum.contiguous()
    dx = torch.empty_like(xn)
    dr = None if (sc is None or not with_res) else torch.empty_like(xn)
    dgb = torch.empty(2, C, dtype=torch.float32, device=xn.device)
    wsb = L.sd_layernorm_workspace_bytes(rows, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=xn.device)
    later = _defer(ctx.pdtype)
    H, W, r = ctx.geom
    rc = L.sd_add_layernorm_patch_bwd(xn.data_ptr(), dy.data_ptr(), None if dp is None else dp.data_ptr(), w.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), None if dres is None else dres.data_ptr(), None if sc is None else sc.data_ptr(), ctx.rps,
                                      dx.data_ptr(), None if dr is None else dr.data_ptr(), None if later else dgb[0].data_ptr(),
                                      None if later else dgb[1].data_ptr(), _DT[xn.dtype], rows, C, H, W, r, ws.data_ptr(), wsb, _stream_ptr())
    _lib.check(rc, 'sd_add_layernorm_patch_bwd')
    if later:
        deferred.add(ws, dgb, 2 * C, L.sd_layernorm_bwd_blocks(rows, C))
    return dx, dr, dgb


class _LayerNormPatchFn(torch.autograd.Function):
    """(y, y_patches) = LayerNorm(x) in token order and in the r x r patch order of the SR conv that reads it."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, H, W, r):
        xc = x.contiguous()
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        _, y, yp, mean, rstd = _patch_fwd(xc, None, None, 1, w, b, eps, H, W, r)
        ctx.save_for_backward(xc, w, mean, rstd)
        ctx.pdtype, ctx.rps, ctx.geom = weight.dtype, 1, (H, W, r)
        ctx.set_materialize_grads(False)
        return y, yp

    @staticmethod
    def backward(ctx, g_y, g_p):
        xc, w, mean, rstd = ctx.saved_tensors
        if g_y is None and g_p is None:
            return (None,) * 7
        dx, _, dgb = _patch_bwd(ctx, xc, w, mean, rstd, None, None, g_y, g_p, False)
        return dx, dgb[0].to(ctx.pdtype), dgb[1].to(ctx.pdtype), None, None, None, None


class _AddLayerNormPatchFn(torch.autograd.Function):
    """(xsum, y, y_patches) = (x + s*res, LayerNorm(x + s*res) in token order and in patch order)."""

    @staticmethod
    def forward(ctx, x, res, scale, weight, bias, eps, H, W, r):
        xc, rc_ = x.contiguous(), res.contiguous()
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        sc = None if scale is None else scale.detach().float().contiguous()
        rps = (xc.numel() // xc.shape[-1]) // xc.shape[0]
        xsum, y, yp, mean, rstd = _patch_fwd(xc, rc_, sc, rps, w, b, eps, H, W, r)
        ctx.save_for_backward(xsum, w, mean, rstd, sc)
        ctx.pdtype, ctx.rps, ctx.geom = weight.dtype, rps, (H, W, r)
        ctx.set_materialize_grads(False)
        return xsum, y, yp

    @staticmethod
    def backward(ctx, g_xsum, g_y, g_p):
        xsum, w, mean, rstd, sc = ctx.saved_tensors
        if g_y is None and g_p is None:
            if g_xsum is None:
                return (None,) * 9
            g_res = g_xsum if sc is None else g_xsum * sc.view(-1, *([1] * (g_xsum.dim() - 1))).to(g_xsum.dtype)
            return (g_xsum, g_res) + (None,) * 7
        dx, dr, dgb = _patch_bwd(ctx, xsum, w, mean, rstd, sc, g_xsum, g_y, g_p, True)
        return dx, (dx if dr is None else dr), None, dgb[0].to(ctx.pdtype), dgb[1].to(ctx.pdtype), None, None, None, None


def layernorm_patches(x, norm, hw, r):
    """norm(x) -> (normed tokens, the same values as [B, (H/r)(W/r), r*r*C] patches)."""
    return _LayerNormPatchFn.apply(x, norm.weight, norm.bias, norm.eps, int(hw[0]), int(hw[1]), int(r))


def add_layernorm_patches(x, res, norm, hw, r, scale=None):
    """add_layernorm with the patch-order copy of the normalised output: -> (xsum, normed, patches)."""
    if res.dtype != x.dtype:
        dt = torch.promote_types(x.dtype, res.dtype)
        x, res = x.to(dt), res.to(dt)
    return _AddLayerNormPatchFn.apply(x, res, scale, norm.weight, norm.bias, norm.eps, int(hw[0]), int(hw[1]), int(r))

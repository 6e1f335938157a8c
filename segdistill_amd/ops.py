"""torch.autograd bindings over the C ABI (device pointers + the current HIP stream)."""
from __future__ import annotations

import ctypes as C
import os

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


class _TokFwdJob(C.Structure):     # include/segdistill_hip.h: sd_cgd_tok_fwd_job
    _fields_ = [('S', C.c_void_p), ('T', C.c_void_p), ('perm', C.c_void_p), ('row_lse2', C.c_void_p), ('row_kl', C.c_void_p), ('loss', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('P', C.c_long), ('B', C.c_int), ('C', C.c_int), ('g', C.c_int),
                ('inv_tau', C.c_float), ('loss_scale', C.c_float), ('reserved', C.c_int)]


class _TokBwdJob(C.Structure):     # sd_cgd_tok_bwd_job
    _fields_ = [('S', C.c_void_p), ('T', C.c_void_p), ('perm', C.c_void_p), ('row_lse2', C.c_void_p), ('upstream', C.c_void_p), ('dS', C.c_void_p),
                ('P', C.c_long), ('B', C.c_int), ('C', C.c_int), ('g', C.c_int), ('inv_tau', C.c_float), ('coef', C.c_float), ('reserved', C.c_int)]


class _CGDKLTokMultiFunction(torch.autograd.Function):
    """n criteria on token-major operands [B, P, C] (csrc/cgd_tok.hip) in ONE call each way: forward = scan launch(es) + one finish launch
    for all of them, backward = one launch.  apply(meta, S_1, T_1, ..., S_n, T_n) -> (loss_1, rows_1, ..., loss_n, rows_n);
    meta = [(group_size, tau, alpha, perm or None)] * n."""

    @staticmethod
    def forward(ctx, meta, *ops_):
        n = len(meta)
        if len(ops_) != 2 * n or n == 0:
            raise ValueError('expected one (S, T) pair per criterion')
        L = _lib.lib()
        if n > L.sd_cgd_kl_tok_max_jobs():
            raise ValueError(f'at most {L.sd_cgd_kl_tok_max_jobs()} criteria per call')
        jobs = (_TokFwdJob * n)()
        keep, outs, saved, info = [], [], [], []
        dt0 = ops_[0].dtype
        for i, (g, tau, alpha, perm) in enumerate(meta):
            S, T = ops_[2 * i], ops_[2 * i + 1]
            _require_gpu(S, T)
            if S.shape != T.shape or S.dim() != 3:
                raise ValueError(f'expected equal token-major [B, P, C] shapes, got {tuple(S.shape)} and {tuple(T.shape)}')
            if S.dtype != T.dtype or S.dtype not in _DT or S.dtype != dt0:
                raise TypeError(f'unsupported dtypes {S.dtype}/{T.dtype} (all criteria of one call share a dtype)')
            S, T = S.contiguous(), T.contiguous()
            B, P, Cc = S.shape
            g = int(g)
            rows = B * (-(-Cc // g))
            if perm is not None:
                perm = perm.to(device=S.device, dtype=torch.int32).contiguous()
                if perm.numel() != Cc:
                    raise ValueError('perm must have C entries')
            ws_bytes = L.sd_cgd_kl_tok_workspace_bytes(B, Cc, P)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=S.device)
            row_lse2 = torch.empty(rows, 2, dtype=torch.float32, device=S.device)
            row_kl = torch.empty(rows, dtype=torch.float32, device=S.device)
            loss = torch.empty((), dtype=torch.float32, device=S.device)
            j = jobs[i]
            j.S, j.T, j.perm, j.row_lse2, j.row_kl, j.loss = S.data_ptr(), T.data_ptr(), _ptr(perm), row_lse2.data_ptr(), row_kl.data_ptr(), loss.data_ptr()
            j.workspace, j.workspace_bytes, j.P, j.B, j.C, j.g = ws.data_ptr(), ws_bytes, P, B, Cc, g
            j.inv_tau, j.loss_scale = 1.0 / float(tau), float(alpha) / rows
            keep.append(ws)
            outs += [loss, row_kl]
            saved += [S, T, row_lse2, perm if perm is not None else torch.empty(0, device=S.device)]
            info.append((g, float(tau), float(alpha), rows, perm is not None))
        _lib.check(L.sd_cgd_kl_tok_fwd_multi(C.cast(jobs, C.c_void_p), n, _DT[dt0], _stream_ptr()), 'sd_cgd_kl_tok_fwd_multi')
        ctx.save_for_backward(*saved)
        ctx.info = info
        ctx.mark_non_differentiable(*outs[1::2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        info, saved = ctx.info, ctx.saved_tensors
        n = len(info)
        jobs = (_TokBwdJob * n)()
        dSs, keep = [], []
        for i, (g, tau, alpha, rows, has_perm) in enumerate(info):
            S, T, row_lse2, perm = saved[4 * i:4 * i + 4]
            B, P, Cc = S.shape
            dS = torch.empty_like(S)
            gl = grads[2 * i]
            up = (gl if gl is not None else torch.zeros((), device=S.device)).to(torch.float32).contiguous()
            j = jobs[i]
            j.S, j.T, j.perm, j.row_lse2, j.upstream, j.dS = S.data_ptr(), T.data_ptr(), perm.data_ptr() if has_perm else None, row_lse2.data_ptr(), up.data_ptr(), dS.data_ptr()
            j.P, j.B, j.C, j.g, j.inv_tau, j.coef = P, B, Cc, g, 1.0 / tau, alpha / (rows * tau)
            dSs.append(dS)
            keep.append(up)
        _lib.check(_lib.lib().sd_cgd_kl_tok_bwd_multi(C.cast(jobs, C.c_void_p), n, _DT[saved[0].dtype], _stream_ptr()), 'sd_cgd_kl_tok_bwd_multi')
        out = [None]
        for dS in dSs:
            out += [dS, None]
        return tuple(out)


def cgd_kl_tokens_supported(S, T, perm_len=None):
    """Token-major [B, P, C] operands the kernels of csrc/cgd_tok.hip take: whole 16-byte channel vectors per pixel."""
    if S.dim() != 3 or S.shape != T.shape or S.dtype != T.dtype or S.dtype not in _DT or not S.is_cuda:
        return False
    n = 4 if S.dtype == torch.float32 else 8
    return S.shape[2] % n == 0 and (perm_len is None or S.shape[2] <= 2048)


def cgd_kl_tokens(S, T, *, group_size, tau, alpha, perm=None, return_rows=False):
    """Channel-group KL on token-major operands [B, P, C] at equal resolution.  HIP only."""
    loss, rows = _CGDKLTokMultiFunction.apply([(group_size, tau, alpha, perm)], S, T)
    return (loss, rows) if return_rows else loss


def cgd_kl_tokens_max_jobs():
    return int(_lib.lib().sd_cgd_kl_tok_max_jobs())


def cgd_kl_tokens_multi(pairs, meta, return_rows=False):
    """Several token-major criteria in one call each way (config 5's four decoder stages): pairs = [(S_i, T_i)], meta = [(group_size, tau,
    alpha, perm or None)].  -> [loss_i] (or [(loss_i, rows_i)])."""
    flat = [x for st in pairs for x in st]
    out = _CGDKLTokMultiFunction.apply(list(meta), *flat)
    return [(out[2 * i], out[2 * i + 1]) if return_rows else out[2 * i] for i in range(len(pairs))]


class _AlignTokJob(C.Structure):   # include/segdistill_hip.h: sd_align_tok_job
    _fields_ = [('X', C.c_void_p), ('W', C.c_void_p), ('bias', C.c_void_p), ('T', C.c_void_p), ('perm', C.c_void_p), ('row_lse2', C.c_void_p),
                ('row_kl', C.c_void_p), ('loss', C.c_void_p), ('upstream', C.c_void_p), ('out', C.c_void_p), ('db_part', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('P', C.c_long), ('B', C.c_int), ('K', C.c_int), ('C', C.c_int),
                ('g', C.c_int), ('inv_tau', C.c_float), ('loss_scale', C.c_float), ('coef', C.c_float), ('reserved', C.c_int)]


_ALIGN_FUSED = os.environ.get('SEGDISTILL_ALIGN_FUSED', '1') == '1'     # A/B: 0 = projection as a token Linear, criterion on its stored output


def align_cgd_tokens_supported(x, weight, t):
    """The fused projection + criterion kernels of csrc/align_tok.hip take this entry: bf16 token-major taps [B, P, Cs] / [B, P, Ct], an fp32
    master weight [Ct, Cs] with Cs in {64, 128, 256} and Ct a multiple of 32."""
    if not (_ALIGN_FUSED and x.is_cuda and x.dim() == 3 and t.dim() == 3 and x.dtype == torch.bfloat16 and t.dtype == torch.bfloat16):
        return False
    if x.shape[:2] != t.shape[:2] or tuple(weight.shape) != (t.shape[2], x.shape[2]) or weight.dtype != torch.float32:
        return False
    return bool(_lib.lib().sd_align_cgd_tok_supported(x.shape[2], t.shape[2]))


class _AlignCGDTokMultiFunction(torch.autograd.Function):
    """n distillation entries `criterion(align(x), t)` on token-major bf16 taps, projection and criterion fused (csrc/align_tok.hip): the projected
    feature is never written.  apply(meta, defer_ok, x_1, W_1, b_1, t_1, ...) -> (loss_1, rows_1, ...); meta = [(group_size, tau, alpha, perm or None)];
    W_i the fp32 master weight [Ct, Cs] (its bf16 copy is the optimizer-maintained shadow), b_i fp32 [Ct] or None.
    Forward: one scan launch + one finish launch for all entries.  Backward: one launch that recomputes the projection and writes dY (bf16) and
    the bias gradient's per-tile column sums, then per entry dX = dY.W and the split-K weight gradient of wgrad_tn.hip."""

    @staticmethod
    def forward(ctx, meta, defer_ok, *ops_):
        from .linear import lowp_copy
        n = len(meta)
        if len(ops_) != 4 * n or n == 0:
            raise ValueError('expected (x, weight, bias, t) per entry')
        L = _lib.lib()
        if n > L.sd_cgd_kl_tok_max_jobs():
            raise ValueError(f'at most {L.sd_cgd_kl_tok_max_jobs()} entries per call')
        jobs = (_AlignTokJob * n)()
        keep, outs, saved, info = [], [], [], []
        for i, (g, tau, alpha, perm) in enumerate(meta):
            x, w, b, t = ops_[4 * i:4 * i + 4]
            _require_gpu(x, t, w)
            if not align_cgd_tokens_supported(x, w, t):
                raise ValueError(f'unsupported operands {tuple(x.shape)} {x.dtype} / {tuple(w.shape)} / {tuple(t.shape)} {t.dtype}')
            x, t = x.contiguous(), t.contiguous()
            wc = lowp_copy(w, torch.bfloat16)
            if not wc.is_contiguous():
                wc = wc.contiguous()
            bb = None if b is None else b.detach().to(torch.float32).contiguous()
            B, P, K = x.shape
            Cc = t.shape[2]
            g = int(g)
            rows = B * (-(-Cc // g))
            if perm is not None:
                perm = perm.to(device=x.device, dtype=torch.int32).contiguous()
                if perm.numel() != Cc:
                    raise ValueError('perm must have C entries')
            ws_bytes = L.sd_align_cgd_tok_workspace_bytes(B, Cc, P)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            row_lse2 = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
            row_kl = torch.empty(rows, dtype=torch.float32, device=x.device)
            loss = torch.empty((), dtype=torch.float32, device=x.device)
            j = jobs[i]
            j.X, j.W, j.bias, j.T, j.perm = x.data_ptr(), wc.data_ptr(), _ptr(bb), t.data_ptr(), _ptr(perm)
            j.row_lse2, j.row_kl, j.loss = row_lse2.data_ptr(), row_kl.data_ptr(), loss.data_ptr()
            j.workspace, j.workspace_bytes, j.P, j.B, j.K, j.C, j.g = ws.data_ptr(), ws_bytes, P, B, K, Cc, g
            j.inv_tau, j.loss_scale = 1.0 / float(tau), float(alpha) / rows
            keep += [ws, bb]
            outs += [loss, row_kl]
            empty = torch.empty(0, device=x.device)
            saved += [x, wc, bb if bb is not None else empty, t, row_lse2, perm if perm is not None else empty]
            info.append((g, float(tau), float(alpha), rows, perm is not None, bb is not None, tuple(w.shape)))
        _lib.check(L.sd_align_cgd_tok_fwd_multi(C.cast(jobs, C.c_void_p), n, _stream_ptr()), 'sd_align_cgd_tok_fwd_multi')
        ctx.save_for_backward(*saved)
        ctx.info, ctx.defer_ok = info, bool(defer_ok)
        ctx.mark_non_differentiable(*outs[1::2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        from . import deferred
        from .linear import linear_bwd_data_bf16, linear_weight_grads
        info, saved = ctx.info, ctx.saved_tensors
        n = len(info)
        L = _lib.lib()
        jobs = (_AlignTokJob * n)()
        work, keep = [], []
        for i, (g, tau, alpha, rows, has_perm, has_bias, w_shape) in enumerate(info):
            x, wc, bb, t, row_lse2, perm = saved[6 * i:6 * i + 6]
            B, P, K = x.shape
            Cc = t.shape[2]
            want_db = has_bias and ctx.needs_input_grad[2 + 4 * i + 2]
            dY = torch.empty_like(t)
            tiles = L.sd_align_cgd_tok_tiles(B, P)
            db_part = torch.empty(tiles, Cc, dtype=torch.float32, device=x.device) if want_db else None
            gl = grads[2 * i]
            up = (gl if gl is not None else torch.zeros((), device=x.device)).to(torch.float32).contiguous()
            j = jobs[i]
            j.X, j.W, j.bias, j.T, j.perm = x.data_ptr(), wc.data_ptr(), bb.data_ptr() if has_bias else None, t.data_ptr(), perm.data_ptr() if has_perm else None
            j.row_lse2, j.upstream, j.out, j.db_part = row_lse2.data_ptr(), up.data_ptr(), dY.data_ptr(), _ptr(db_part)
            j.P, j.B, j.K, j.C, j.g, j.inv_tau, j.coef = P, B, K, Cc, g, 1.0 / tau, alpha / (rows * tau)
            work.append((x, wc, dY, db_part, tiles, w_shape))
            keep.append(up)
        _lib.check(L.sd_align_cgd_tok_bwd_multi(C.cast(jobs, C.c_void_p), n, _stream_ptr()), 'sd_align_cgd_tok_bwd_multi')
        out = [None, None]
        for i, (x, wc, dY, db_part, tiles, w_shape) in enumerate(work):
            need = ctx.needs_input_grad[2 + 4 * i:2 + 4 * i + 4]
            dy2 = dY.view(-1, dY.shape[-1])
            dx = linear_bwd_data_bf16(dy2, wc).view(x.shape) if need[0] else None
            dw = linear_weight_grads(x, dy2, w_shape, torch.float32, False, ctx.defer_ok, ctx.defer_ok)[0] if need[1] else None
            db = None
            if db_part is not None:
                Cc = dY.shape[-1]
                buf = torch.empty(Cc, dtype=torch.float32, device=x.device)
                if ctx.defer_ok and deferred.enabled():
                    deferred.add(db_part, buf, Cc, tiles)
                else:
                    deferred.reduce_now(db_part, buf, Cc, tiles)
                db = buf.view(Cc)
            out += [dx, dw, db, None]
        return tuple(out)


def align_cgd_tokens_multi(entries, meta, defer_ok=True, return_rows=False):
    """Several fused (align + criterion) entries in one call each way: entries = [(x_i, weight_i, bias_i, t_i)], meta = [(group_size, tau, alpha,
    perm or None)] -> [loss_i] (or [(loss_i, rows_i)]).  All entries of a call share the student width Cs."""
    flat = [v for e in entries for v in e]
    out = _AlignCGDTokMultiFunction.apply(list(meta), defer_ok, *flat)
    return [(out[2 * i], out[2 * i + 1]) if return_rows else out[2 * i] for i in range(len(entries))]


# ------------------------------------------------------------------------------------------------
def can_fuse_resize(x_student, x_teacher, out_size, transform_config=None) -> bool:
    """True when the fused-upsample CGD kernels cover this case (same tap shapes, integer factor
    2/4/8 on both axes, 'channel' rows)."""
    if transform_config is None or transform_config.get('loss_type') != 'channel':
        return False
    if x_student.dim() != 4 or x_student.shape != x_teacher.shape or x_student.dtype != x_teacher.dtype:
        return False
    if x_student.dtype not in _DT or not x_student.is_cuda:
        return False
    h, w = x_student.shape[2:]
    return bool(_lib.lib().sd_cgd_kl_up_supported(int(h), int(w), int(out_size[0]), int(out_size[1])))


class _CGDKLUpFunction(torch.autograd.Function):
    """CGD/CD criterion on TAP tensors with the bilinear resize to `out_size` fused in (R2)."""

    @staticmethod
    def forward(ctx, s, t, out_size, group_size, tau, alpha, perm):
        _require_gpu(s, t)
        if s.shape != t.shape or s.dim() != 4:
            raise ValueError(f'expected equal 4-D shapes, got {tuple(s.shape)} and {tuple(t.shape)}')
        if s.dtype != t.dtype or s.dtype not in _DT:
            raise TypeError(f'unsupported dtypes {s.dtype}/{t.dtype}')
        s, t = s.contiguous(), t.contiguous()
        B, Cc, h, w = s.shape
        H, W = int(out_size[0]), int(out_size[1])
        g = int(group_size)
        rows = B * (-(-Cc // g))
        L = _lib.lib()
        if perm is not None:
            perm = perm.to(device=s.device, dtype=torch.int32).contiguous()
            if perm.numel() != Cc:
                raise ValueError('perm must have C entries')
        ws_bytes = L.sd_cgd_kl_up_workspace_bytes(B, Cc, h, w, H, W, g)
        if ws_bytes == 0:
            raise RuntimeError(f'no fused-upsample kernel for {h}x{w} -> {H}x{W}')
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=s.device)
        row_lse2 = torch.empty(rows, 2, dtype=torch.float32, device=s.device)
        row_kl = torch.empty(rows, dtype=torch.float32, device=s.device)
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        rc = L.sd_cgd_kl_up_fwd(s.data_ptr(), t.data_ptr(), _DT[s.dtype], B, Cc, h, w, H, W, g, 1.0 / float(tau),
                                float(alpha) / rows, _ptr(perm), row_lse2.data_ptr(), row_kl.data_ptr(), loss.data_ptr(),
                                ws.data_ptr(), ws_bytes, _stream_ptr())
        _lib.check(rc, 'sd_cgd_kl_up_fwd')
        ctx.save_for_backward(s, t, row_lse2, perm if perm is not None else torch.empty(0, device=s.device))
        ctx.meta = (H, W, g, float(tau), float(alpha), rows, perm is not None)
        ctx.mark_non_differentiable(row_kl)
        return loss, row_kl

    @staticmethod
    def backward(ctx, grad_loss, _grad_rows):
        s, t, row_lse2, perm = ctx.saved_tensors
        H, W, g, tau, alpha, rows, has_perm = ctx.meta
        B, Cc, h, w = s.shape
        ds = torch.empty_like(s)
        up = grad_loss.to(torch.float32).contiguous()
        rc = _lib.lib().sd_cgd_kl_up_bwd(s.data_ptr(), t.data_ptr(), _DT[s.dtype], B, Cc, h, w, H, W, g, 1.0 / tau,
                                         alpha / (rows * tau), perm.data_ptr() if has_perm else None, row_lse2.data_ptr(),
                                         up.data_ptr(), ds.data_ptr(), _stream_ptr())
        _lib.check(rc, 'sd_cgd_kl_up_bwd')
        return ds, None, None, None, None, None, None


class _CGDKLUp2Function(torch.autograd.Function):
    """Two channel criteria (a, b) on the SAME taps with the bilinear resize fused in, one pass each way (csrc/cgd_up.hip DUAL kernels):
    apply(s, t, out_size, (g_a, tau_a, alpha_a), (g_b, tau_b, alpha_b), perm) -> (loss_a, rows_a, loss_b, rows_b); `perm` orders the channel
    slots of both (callers fuse only when that is right: see include/segdistill_hip.h, sd_cgd_kl_up_fwd2)."""

    @staticmethod
    def forward(ctx, s, t, out_size, crit_a, crit_b, perm):
        _require_gpu(s, t)
        if s.shape != t.shape or s.dim() != 4:
            raise ValueError(f'expected equal 4-D shapes, got {tuple(s.shape)} and {tuple(t.shape)}')
        if s.dtype != t.dtype or s.dtype not in _DT:
            raise TypeError(f'unsupported dtypes {s.dtype}/{t.dtype}')
        s, t = s.contiguous(), t.contiguous()
        B, Cc, h, w = s.shape
        H, W = int(out_size[0]), int(out_size[1])
        (ga, taua, alphaa), (gb, taub, alphab) = crit_a, crit_b
        ga, gb = int(ga), int(gb)
        rows_a, rows_b = B * (-(-Cc // ga)), B * (-(-Cc // gb))
        L = _lib.lib()
        if perm is not None:
            perm = perm.to(device=s.device, dtype=torch.int32).contiguous()
            if perm.numel() != Cc:
                raise ValueError('perm must have C entries')
        ws_bytes = 2 * L.sd_cgd_kl_up_workspace_bytes(B, Cc, h, w, H, W, ga)
        if ws_bytes == 0:
            raise RuntimeError(f'no fused-upsample kernel for {h}x{w} -> {H}x{W}')
        f32 = dict(dtype=torch.float32, device=s.device)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=s.device)
        lse_a, kl_a, loss_a = torch.empty(rows_a, 2, **f32), torch.empty(rows_a, **f32), torch.empty((), **f32)
        lse_b, kl_b, loss_b = torch.empty(rows_b, 2, **f32), torch.empty(rows_b, **f32), torch.empty((), **f32)
        rc = L.sd_cgd_kl_up_fwd2(s.data_ptr(), t.data_ptr(), _DT[s.dtype], B, Cc, h, w, H, W, _ptr(perm),
                                 ga, 1.0 / float(taua), float(alphaa) / rows_a, lse_a.data_ptr(), kl_a.data_ptr(), loss_a.data_ptr(),
                                 gb, 1.0 / float(taub), float(alphab) / rows_b, lse_b.data_ptr(), kl_b.data_ptr(), loss_b.data_ptr(),
                                 ws.data_ptr(), ws_bytes, _stream_ptr())
        _lib.check(rc, 'sd_cgd_kl_up_fwd2')
        ctx.save_for_backward(s, t, lse_a, lse_b, perm if perm is not None else torch.empty(0, device=s.device))
        ctx.meta = (H, W, (ga, float(taua), float(alphaa), rows_a), (gb, float(taub), float(alphab), rows_b), perm is not None)
        ctx.mark_non_differentiable(kl_a, kl_b)
        return loss_a, kl_a, loss_b, kl_b

    @staticmethod
    def backward(ctx, grad_a, _ra, grad_b, _rb):
        s, t, lse_a, lse_b, perm = ctx.saved_tensors
        H, W, (ga, taua, alphaa, rows_a), (gb, taub, alphab, rows_b), has_perm = ctx.meta
        B, Cc, h, w = s.shape
        ds = torch.empty_like(s)
        zero = None
        ups = []
        for gr in (grad_a, grad_b):
            if gr is None:
                zero = torch.zeros((), dtype=torch.float32, device=s.device) if zero is None else zero
                gr = zero
            ups.append(gr.to(torch.float32).contiguous())
        rc = _lib.lib().sd_cgd_kl_up_bwd2(s.data_ptr(), t.data_ptr(), _DT[s.dtype], B, Cc, h, w, H, W, perm.data_ptr() if has_perm else None,
                                          ga, 1.0 / taua, alphaa / (rows_a * taua), lse_a.data_ptr(), ups[0].data_ptr(),
                                          gb, 1.0 / taub, alphab / (rows_b * taub), lse_b.data_ptr(), ups[1].data_ptr(), ds.data_ptr(), _stream_ptr())
        _lib.check(rc, 'sd_cgd_kl_up_bwd2')
        return ds, None, None, None, None, None


def cgd_kl_up2(s, t, out_size, crit_a, crit_b, perm=None, return_rows=False):
    """crit_x = (group_size, tau, alpha).  -> (loss_a, loss_b) or ((loss_a, rows_a), (loss_b, rows_b))."""
    la, ra, lb, rb = _CGDKLUp2Function.apply(s, t, tuple(out_size), tuple(crit_a), tuple(crit_b), perm)
    return ((la, ra), (lb, rb)) if return_rows else (la, lb)


def cgd_kl_up(s, t, out_size, *, group_size, tau, alpha, perm=None, return_rows=False):
    loss, rows = _CGDKLUpFunction.apply(s, t, tuple(out_size), group_size, tau, alpha, perm)
    return (loss, rows) if return_rows else loss


class _PixKLFunction(torch.autograd.Function):
    """loss = alpha/(B*H*W) * sum_pixels KL(softmax_C(T/tau) || softmax_C(S/tau))."""

    @staticmethod
    def forward(ctx, S, T, tau, alpha):
        _require_gpu(S, T)
        if S.shape != T.shape or S.dim() != 4:
            raise ValueError(f'expected equal 4-D shapes, got {tuple(S.shape)} and {tuple(T.shape)}')
        if S.dtype != T.dtype or S.dtype not in _DT:
            raise TypeError(f'unsupported dtypes {S.dtype}/{T.dtype}')
        S, T = S.contiguous(), T.contiguous()
        B, Cc, H, W = S.shape
        rows = B * H * W
        L = _lib.lib()
        ws_bytes = L.sd_pix_kl_workspace_bytes(B, Cc, H, W)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=S.device)
        lse2 = torch.empty(2, rows, dtype=torch.float32, device=S.device)
        loss = torch.empty((), dtype=torch.float32, device=S.device)
        rc = L.sd_pix_kl_fwd(S.data_ptr(), T.data_ptr(), _DT[S.dtype], B, Cc, H, W, 1.0 / float(tau), float(alpha) / rows,
                             lse2.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws_bytes, _stream_ptr())
        _lib.check(rc, 'sd_pix_kl_fwd')
        ctx.save_for_backward(S, T, lse2)
        ctx.meta = (float(tau), float(alpha), rows)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        S, T, lse2 = ctx.saved_tensors
        tau, alpha, rows = ctx.meta
        B, Cc, H, W = S.shape
        dS = torch.empty_like(S)
        up = grad_loss.to(torch.float32).contiguous()
        rc = _lib.lib().sd_pix_kl_bwd(S.data_ptr(), T.data_ptr(), _DT[S.dtype], B, Cc, H, W, 1.0 / tau, alpha / (rows * tau),
                                      lse2.data_ptr(), up.data_ptr(), dS.data_ptr(), _stream_ptr())
        _lib.check(rc, 'sd_pix_kl_bwd')
        return dS, None, None, None


def pix_kl(S, T, *, tau, alpha):
    return _PixKLFunction.apply(S, T, tau, alpha)


def can_fuse_pixel_resize(x_student, x_teacher, out_size) -> bool:
    """The fused-upsample pixel-wise kernels (csrc/pix_up.hip) cover this case: same tap shapes, integer factor 2 / 4 / 8 on both axes."""
    if x_student.dim() != 4 or x_student.shape != x_teacher.shape or x_student.dtype != x_teacher.dtype:
        return False
    if x_student.dtype not in _DT or not x_student.is_cuda:
        return False
    h, w = x_student.shape[2:]
    return bool(_lib.lib().sd_pix_kl_up_supported(int(h), int(w), int(out_size[0]), int(out_size[1])))


class _PixKLUpFunction(torch.autograd.Function):
    """Pixel-wise criterion (softmax over the classes of every pixel) on TAP tensors with the bilinear resize to `out_size` fused in."""

    @staticmethod
    def forward(ctx, s, t, out_size, tau, alpha):
        _require_gpu(s, t)
        if s.shape != t.shape or s.dim() != 4 or s.dtype != t.dtype or s.dtype not in _DT:
            raise ValueError(f'expected equal 4-D taps of one supported dtype, got {tuple(s.shape)} {s.dtype} / {tuple(t.shape)} {t.dtype}')
        s, t = s.contiguous(), t.contiguous()
        B, Cc, h, w = s.shape
        H, W = int(out_size[0]), int(out_size[1])
        rows = B * H * W
        L = _lib.lib()
        wsb = L.sd_pix_kl_up_workspace_bytes(B, h)
        ws = torch.empty(wsb, dtype=torch.uint8, device=s.device)
        lse2 = torch.empty(2, rows, dtype=torch.float32, device=s.device)
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        _lib.check(L.sd_pix_kl_up_fwd(s.data_ptr(), t.data_ptr(), _DT[s.dtype], B, Cc, h, w, H, W, 1.0 / float(tau), float(alpha) / rows,
                                      lse2.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, _stream_ptr()), 'sd_pix_kl_up_fwd')
        ctx.save_for_backward(s, t, lse2)
        ctx.meta = (H, W, float(tau), float(alpha), rows)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        s, t, lse2 = ctx.saved_tensors
        H, W, tau, alpha, rows = ctx.meta
        B, Cc, h, w = s.shape
        ds = torch.empty_like(s)
        up = grad_loss.to(torch.float32).contiguous()
        _lib.check(_lib.lib().sd_pix_kl_up_bwd(s.data_ptr(), t.data_ptr(), _DT[s.dtype], B, Cc, h, w, H, W, 1.0 / tau, alpha / (rows * tau),
                                               lse2.data_ptr(), up.data_ptr(), ds.data_ptr(), _stream_ptr()), 'sd_pix_kl_up_bwd')
        return ds, None, None, None, None


def pix_kl_up(s, t, out_size, *, tau, alpha):
    """PDLoss on the taps: class softmax at label resolution, nothing of label size materialised.  HIP only."""
    return _PixKLUpFunction.apply(s, t, out_size, tau, alpha)


class _ATKLFunction(torch.autograd.Function):
    """ATLoss: mean_{b,p}(mean_c S - mean_c T)^2 + 1/(B*H*W) * sum_pixels KL(softmax_C(T) || softmax_C(S)), one pass each way."""

    @staticmethod
    def forward(ctx, S, T):
        _require_gpu(S, T)
        if S.shape != T.shape or S.dim() != 4:
            raise ValueError(f'expected equal 4-D shapes, got {tuple(S.shape)} and {tuple(T.shape)}')
        if S.dtype != T.dtype or S.dtype not in _DT:
            raise TypeError(f'unsupported dtypes {S.dtype}/{T.dtype}')
        S, T = S.contiguous(), T.contiguous()
        B, Cc, H, W = S.shape
        L = _lib.lib()
        ws_bytes = L.sd_pix_kl_workspace_bytes(B, Cc, H, W)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=S.device)
        planes = torch.empty(3, B * H * W, dtype=torch.float32, device=S.device)
        loss = torch.empty((), dtype=torch.float32, device=S.device)
        rc = L.sd_at_kl_fwd(S.data_ptr(), T.data_ptr(), _DT[S.dtype], B, Cc, H, W, planes.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws_bytes,
                            _stream_ptr())
        _lib.check(rc, 'sd_at_kl_fwd')
        ctx.save_for_backward(S, T, planes)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        S, T, planes = ctx.saved_tensors
        B, Cc, H, W = S.shape
        dS = torch.empty_like(S)
        up = grad_loss.to(torch.float32).contiguous()
        rc = _lib.lib().sd_at_kl_bwd(S.data_ptr(), T.data_ptr(), _DT[S.dtype], B, Cc, H, W, planes.data_ptr(), up.data_ptr(), dS.data_ptr(),
                                     _stream_ptr())
        _lib.check(rc, 'sd_at_kl_bwd')
        return dS, None


def at_kl(S, T):
    return _ATKLFunction.apply(S, T)


class _IFVDFunction(torch.autograd.Function):
    """10 * mean_p (cos(S_p, centre^S_label(p)) - cos(T_p, centre^T_label(p)))^2 with class centres = per-image class means."""

    @staticmethod
    def forward(ctx, S, T, cls, n_cls):
        _require_gpu(S, T)
        if S.shape != T.shape or S.dim() != 4:
            raise ValueError(f'expected equal 4-D shapes, got {tuple(S.shape)} and {tuple(T.shape)}')
        if S.dtype != T.dtype or S.dtype not in _DT:
            raise TypeError(f'unsupported dtypes {S.dtype}/{T.dtype}')
        S, T = S.contiguous(), T.contiguous()
        B, Cc, H, W = S.shape
        HW, K = H * W, int(n_cls)
        cls = cls.reshape(B, HW).to(torch.int32).contiguous()
        L, dt, st = _lib.lib(), _DT[S.dtype], _stream_ptr()
        f32 = dict(dtype=torch.float32, device=S.device)
        counts = torch.empty(B, K, dtype=torch.int32, device=S.device)
        smask = torch.empty(L.sd_ifvd_stepmask_ints(B, HW, K), dtype=torch.int32, device=S.device)
        _lib.check(L.sd_ifvd_counts(cls.data_ptr(), B, HW, K, counts.data_ptr(), smask.data_ptr(), st), 'sd_ifvd_counts')
        wsb = L.sd_ifvd_workspace_bytes(B, Cc, HW, K)
        ws = torch.empty(wsb, dtype=torch.uint8, device=S.device)
        mean_s, mean_t = torch.empty(B, Cc, K, **f32), torch.empty(B, Cc, K, **f32)         # [B][C][K] (csrc/ifvd.hip)
        _lib.check(L.sd_ifvd_class_means(S.data_ptr(), T.data_ptr(), dt, cls.data_ptr(), smask.data_ptr(), counts.data_ptr(), mean_s.data_ptr(),
                                         mean_t.data_ptr(), ws.data_ptr(), wsb, B, Cc, HW, K, st), 'sd_ifvd_class_means')
        coefs = torch.empty(3, B * HW, **f32)
        loss = torch.empty((), **f32)
        _lib.check(L.sd_ifvd_cos(S.data_ptr(), T.data_ptr(), dt, cls.data_ptr(), mean_s.data_ptr(), mean_t.data_ptr(), coefs.data_ptr(), loss.data_ptr(),
                                 ws.data_ptr(), wsb, B, Cc, HW, K, st), 'sd_ifvd_cos')
        ctx.save_for_backward(S, cls, smask, counts, mean_s, coefs)
        ctx.K = K
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        S, cls, smask, counts, mean_s, coefs = ctx.saved_tensors
        B, Cc, H, W = S.shape
        HW, K = H * W, ctx.K
        L, dt, st = _lib.lib(), _DT[S.dtype], _stream_ptr()
        f32 = dict(dtype=torch.float32, device=S.device)
        A, Bk = torch.empty(B, Cc, K, **f32), torch.empty(B, K, **f32)
        wsb = L.sd_ifvd_workspace_bytes(B, Cc, HW, K)
        ws = torch.empty(wsb, dtype=torch.uint8, device=S.device)
        _lib.check(L.sd_ifvd_coef_sums(S.data_ptr(), dt, cls.data_ptr(), smask.data_ptr(), counts.data_ptr(), coefs.data_ptr(), A.data_ptr(), Bk.data_ptr(),
                                       ws.data_ptr(), wsb, B, Cc, HW, K, st), 'sd_ifvd_coef_sums')
        dS = torch.empty_like(S)
        up = grad_loss.to(torch.float32).contiguous()
        _lib.check(L.sd_ifvd_bwd(S.data_ptr(), dt, cls.data_ptr(), mean_s.data_ptr(), coefs.data_ptr(), A.data_ptr(), Bk.data_ptr(), counts.data_ptr(),
                                 up.data_ptr(), dS.data_ptr(), B, Cc, HW, K, st), 'sd_ifvd_bwd')
        return dS, None, None, None


def ifvd_term(S, T, cls, n_cls):
    """cls: integer class per pixel [B,H,W] (or [B,1,H,W]); values outside [0, n_cls) mean "no class"."""
    return _IFVDFunction.apply(S, T, cls, n_cls)


def align1x1(x, weight, bias=None):
    """Y[b,:,p] = W[Ct,Cs] . X[b,:,p] + bias  (MFMA GEMM kernel; see csrc/align1x1.hip)."""
    from .align import align1x1 as _impl
    return _impl(x, weight, bias)

"""nn.Linear on token-major activations.  Forward and input gradient: the measured three-way dispatch of `_gemm_mode` below (csrc/token_gemm.hip
in split-bf16 or exact-f32 arithmetic, or the library GEMM); weight gradient dW = dY^T . X -- tall-skinny at the high-resolution MiT stages,
where the library kernels ran 30x off the HBM roofline (profiles/r01_train_step_kernels_*.txt) -- by the split-K MFMA kernel
(csrc/align1x1.hip: sd_linear_wgrad) with its slabs combined by the deferred batched reduction; the class-plane `linear_pred`
(sd_linear_nchw_*) and the optional long-K / patch forms live here too."""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import _lib, deferred, planes, token_gemm
from .ops import _DT, _stream_ptr

MIN_TOKENS = 1     # every training Linear: below 8192 tokens dW is the library's GEMM, but the bias gradient still avoids ATen's
                   # memset + multi-block sum(0) pair (deferred.column_sum)
# A/B on MI355X (config 2, same box, 30 steps): 19.9 ms/step with the split-K kernel vs 19.4 ms with the library GEMM -- the
# library's un-split 64-workgroup kernel is slow in isolation (183 us) but leaves the chip to the concurrently running
# student/teacher stream, while the split-K version occupies all CUs.  Kept as an opt-in (SEGDISTILL_LONGK=1).
_LONGK_ENABLED = os.environ.get('SEGDISTILL_LONGK') == '1'
# The frozen network's SR conv with the patch gather inside the GEMM (patch_linear_forward) removes the [rows, r*r*C] copy, but as a split-K
# kernel it hits the same wall: A/B on MI355X (config 2, same box, 30 steps, twice): 659.5 / 662.8 imgs/s with it, 670.8 / 671.3 without.
# Opt-in (SEGDISTILL_PATCH_GEMM=1).
_PATCH_GEMM = os.environ.get('SEGDISTILL_PATCH_GEMM') == '1'


# Forward and input gradient of fp32 Linears: the library GEMM, or the kernels of csrc/token_gemm.hip in one of two arithmetic modes.
# `_gemm_mode` is the MEASURED dispatch (tools/gemm_bench.py on MI355X, device time inside a replayed graph; profiles/r02_gemm_bench.txt):
#   'x3'   split-bf16 products on the bf16 matrix pipe (fp32-level accuracy: every fp32 operand is split exactly into three bf16 terms, six
#          products kept; tests/test_token_gemm_gpu.py holds it to the f32 path's error bound).  12 matrix-pipe cycles per k instead of 32:
#          ahead of the library by 14-26 % on the wide products that fill the chip with 128 x 128 tiles (head fuse 256 -> 256: 119 vs 139 us,
#          teacher fc1 320 -> 1280: 52 vs 61 us, 32 -> 256: 31.6 vs 42.8 us);
#   'f32'  v_mfma_f32_32x32x2_f32, bit-equal to an fmaf chain: matches the library on the MFMA-bound shapes (76 % vs 79 % of the f32-input peak)
#          and is ahead only on the short-reduction products of stages 1-2 (K <= 64 over >= 32768 tokens: 9.7 vs 12.8 us for 32 -> 32);
#   'lib'  everything else (few tokens, long reductions: the library's split-K / small-tile kernels).
# SEGDISTILL_TOKEN_GEMM=0 forces the library everywhere, SEGDISTILL_SPLIT_BF16=0 keeps the kernels on exact-f32 arithmetic only (A/B runs).
_TOKEN_GEMM = os.environ.get('SEGDISTILL_TOKEN_GEMM', '1') == '1'
_SPLIT_BF16 = os.environ.get('SEGDISTILL_SPLIT_BF16', '1') == '1'


def _gemm_mode(tokens, k, n):
    if not _TOKEN_GEMM:
        return 'lib'
    tiles = -(-tokens // 128) * -(-n // 128)
    # 192 tiles of 128 x 128 fill the chip; from 128 tiles on, the 64-row tile of the planes kernel (csrc: dispatch_planes) is ahead of the
    # library for reductions up to 512 (2048 x 256 -> 1024: 11.4 vs 14.1 us, 8192 x 256 -> 256: 11.6 vs 13.3, 2048 x 512 -> 1024: 18.5 vs 21.2)
    if _SPLIT_BF16 and n >= 128 and k % 32 == 0 and n % 4 == 0 and (tiles >= 192 or (tiles >= 128 and k <= 512)) and \
            (-(-n // 128) * 128) * 4 <= n * 5:
        return 'x3'
    if tokens >= 32768 and k <= 64:
        return 'f32'
    return 'lib'


def linear_forward(x, weight, bias):
    """x . W^T + bias without autograd bookkeeping (frozen networks, and the forward of _TokenLinear): the measured three-way dispatch."""
    if x.dtype == torch.float32 and token_gemm.supported(x, weight):
        mode = _gemm_mode(x.numel() // x.shape[-1], x.shape[-1], weight.shape[0])
        if mode == 'x3' and planes.supported(weight):
            # the weight's bf16 planes are split ONCE per optimizer step (never, for a frozen weight), not in every k-step (planes.py)
            return token_gemm.linear_fwd_planes(x, weight, planes.get(weight, 'fwd'), bias)
        if mode != 'lib':
            return token_gemm.linear_fwd(x, weight, bias, split_bf16=(mode == 'x3'))
    if x.is_cuda and weight.dtype == torch.float32 and not weight.requires_grad and (bias is None or not bias.requires_grad):
        # low-precision storage (autocast): a FROZEN fp32 weight would be cast again on every call -- 644 cast kernels, 2.07 ms of a 14.1 ms
        # config-5 step for the 41-layer B4 teacher.  The cast copy is cached per parameter (frozen_derived: invalidated by a checkpoint load).
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else x.dtype
        if dt in (torch.bfloat16, torch.float16):
            from .layers import frozen_derived
            # cached on the BASE parameter, the view's geometry in the key: the SR conv's patch-ordered matrix and the head's linear_fuse
            # blocks are VIEWS made anew on every call, and a cache keyed on the view object never hit -- 42 cast kernels per config-5 step
            # for the B4 teacher's 38 SR convs and 4 fuse blocks (tools/cast_probe.py, round 4)
            root = weight._base if weight._base is not None else weight
            if root.requires_grad:
                w16 = weight.to(dt)
            else:
                w16 = frozen_derived(root, ('cast', dt, weight.data_ptr(), tuple(weight.shape), tuple(weight.stride())), lambda: weight.to(dt))
            b16 = None if bias is None else frozen_derived(bias, ('cast', dt), lambda: bias.to(dt))
            return linear_fwd_bf16(x if x.dtype == dt else x.to(dt), w16, b16) if dt == torch.bfloat16 else F.linear(x if x.dtype == dt else x.to(dt), w16, b16)
    return F.linear(x, weight, bias)


def _bwd_data(dy2, weight):
    """dy2 [T, N] . W [N, K] -> [T, K] (None: not ours, use the library)."""
    if dy2.dtype == torch.float32 and weight.dtype == torch.float32 and dy2.is_cuda and weight.dim() == 2:
        mode = _gemm_mode(dy2.shape[0], weight.shape[0], weight.shape[1])
        if mode == 'x3' and planes.supported(weight):
            return token_gemm.linear_bwd_data_planes(dy2, weight, planes.get(weight, 'bwd'))
        if mode != 'lib':
            return token_gemm.linear_bwd_data(dy2, weight, split_bf16=(mode == 'x3'))
    return None


# bf16 Linears (config 5): csrc/tok_gemm_bf16.hip instead of the library wherever its shape test passes; SEGDISTILL_BF16_TOK_GEMM=0: A/B switch.
_BF16_TOK_GEMM = os.environ.get('SEGDISTILL_BF16_TOK_GEMM', '1') == '1'


def bf16_tok_gemm_ok(tokens, k, n):
    """The MEASURED dispatch of the bf16 forward product (tools/bf16_gemm_bench.py on MI355X; profiles/r05_bf16_gemm_bench.txt)."""
    # ours: the latency-bound products (a few GFLOP over <= 32768 tokens, or short reductions over the high-resolution stages) -- 8192 x 320 -> 320:
    # 6.7 vs 9.3 us, -> 1280: 14.3 vs 19.0, 2048 x 320 -> 640: 5.7 vs 7.7, 131072 x 64 -> 64: 8.8 vs 11.8; the library: the throughput-bound ones
    # (131072 x 768 -> 768: 137 vs 221 us -- 128 x 64 tiles are LDS-fill-bound there) and long reductions over few tokens (2048 x 2048 -> 512: its
    # split-K kernels, 12 vs 15 us)
    if not _BF16_TOK_GEMM or 2.0 * tokens * k * n >= 9e9 or (k > 1024 and tokens < 8192):
        return False
    return bool(_lib.lib().sd_linear_bf16_fwd_supported(tokens, k, n))


def linear_fwd_bf16(x, w, b=None):
    """F.linear(x, w, b) for bf16 x [..., in] and a row-major bf16 w [out, in] (b: bf16 or fp32): fp32 accumulation, one rounding."""
    K, N = x.shape[-1], w.shape[0]
    if (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.dim() == 2 and w.stride() == (K, 1) and x.is_contiguous()
            and (b is None or (b.dtype in _DT and b.is_contiguous())) and x.numel() > 0 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0):
        T = x.numel() // K
        if bf16_tok_gemm_ok(T, K, N):
            y = torch.empty(*x.shape[:-1], N, dtype=torch.bfloat16, device=x.device)
            _lib.check(_lib.lib().sd_linear_bf16_fwd(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), 0 if b is None else _DT[b.dtype],
                                                     y.data_ptr(), T, K, N, _stream_ptr()), 'sd_linear_bf16_fwd')
            return y
    return F.linear(x, w, b if b is None or b.dtype == x.dtype else b.to(x.dtype))


# Stand-alone the library is ahead at 256 input features (62.0 vs 73.6 us over 131072 tokens, profiles/r06_tok_dx_bench.txt), but INSIDE the config-5
# step the kernel is not behind (753.0 with it vs 746.8 / 749.1 imgs/s with the library, same box, profiles/r06_ab_cfg5_tok_dx.txt): it stays.
_TOK_DX_MAX_IN = int(os.environ.get('SEGDISTILL_TOK_DX_MAX_IN', '256'))     # A/B: 128 = the library for config 5's projection shape


def linear_bwd_data_bf16(dy2, wc):
    """dX [T, in] = dY [T, out] . W [out, in] for bf16 operands (the align projection's input gradient).  MEASURED dispatch
    (tools/tok_dx_bench.py on MI355X, profiles/r06_tok_dx_bench.txt): csrc/align_tok.hip's tok_dx_kernel for the projection's shapes (768 -> 64:
    13-39 us against the library's 23-42, 768 -> 128: 18-47 vs 23-47; 768 -> 256: 73.6 vs 62.0 stand-alone, level inside the step), the library
    otherwise."""
    T, M = dy2.shape
    N = wc.shape[1]
    L = _lib.lib()
    if (_TOK_DX_MAX_IN >= N and dy2.is_cuda and dy2.dtype == torch.bfloat16 and wc.dtype == torch.bfloat16 and dy2.is_contiguous() and wc.is_contiguous()
            and L.sd_align_cgd_tok_supported(N, M) and dy2.data_ptr() % 16 == 0 and wc.data_ptr() % 16 == 0):
        dx = torch.empty(T, N, dtype=torch.bfloat16, device=dy2.device)
        _lib.check(L.sd_linear_tok_bf16_bwd_data(dy2.data_ptr(), wc.data_ptr(), dx.data_ptr(), T, M, N, _stream_ptr()), 'sd_linear_tok_bf16_bwd_data')
        return dx
    return dy2 @ wc


_BF16_WGRAD_LIB = os.environ.get('SEGDISTILL_BF16_WGRAD_LIB', '0') == '1'
# A/B: 0 = the bias gradient of the transposed-read fp32 weight gradients from the batched column-sum pass (a second read of dY) instead of riding along
# in their slabs.  Config 2, same box: 793.2 - 795.5 imgs/s with it, 787.0 / 787.1 without (profiles/r04_ab_cfg2_tn_fused_bias.txt)
_TN_FUSED_BIAS = os.environ.get('SEGDISTILL_TN_FUSED_BIAS', '1') == '1'
_SPLITK_WGRAD = os.environ.get('SEGDISTILL_SPLITK_WGRAD', '1') == '1'      # A/B: 0 = the library's dY^T @ X for the non-tall-skinny weight gradients
# A/B: 0 = the fp32 weight gradients launched one by one inside the backward (rounds 2-4) instead of the scope's grouped launches (deferred.add_wgrad)
_FP32_WGRAD_GROUPED = os.environ.get('SEGDISTILL_FP32_WGRAD_GROUPED', '1') == '1'


def lowp_copy(t, dt):
    """t.to(dt) for a low-precision forward.  For a TRAINABLE fp32 parameter under bf16 storage the copy is kept on the parameter as its
    shadow (`_sd_shadow` = (parameter version, tensor)): engine/optim.py::HipAdamW rewrites it together with the parameter, so the next
    forward -- eager or a graph replay -- finds the cast already done instead of running one tiny kernel per weight and bias.  Any torch
    op that writes the parameter (checkpoint load, another optimizer) bumps its version and the shadow is remade.  (A write through
    `param.data` -- or by a torch FUSED optimizer -- does NOT bump it: engine/optim.py's step post-hook rewrites the shadows then; other code
    that edits weights that way mid-training must delete `param._sd_shadow`.)"""
    base = t._base
    if (dt == torch.bfloat16 and base is not None and isinstance(base, torch.nn.Parameter) and base.requires_grad and base.dtype == torch.float32
            and base.is_cuda and t.dtype == torch.float32):
        # a VIEW of a trainable parameter (the SR conv's patch-ordered matrix, a block of linear_fuse): the same view of the parameter's shadow --
        # the shadow keeps the parameter's layout, so shape / strides / offset carry over -- instead of one cast kernel per view and step
        sh = lowp_copy(base, dt)
        if sh.stride() == base.stride():
            return sh.as_strided(t.shape, t.stride(), t.storage_offset() - base.storage_offset() + sh.storage_offset())
    if not (dt == torch.bfloat16 and isinstance(t, torch.nn.Parameter) and t.requires_grad and t.dtype == torch.float32 and t.is_cuda):
        return t.to(dt)
    sh = getattr(t, '_sd_shadow', None)
    if sh is not None and sh[0] == t._version and sh[1].dtype == dt:
        return sh[1]
    if torch.cuda.is_current_stream_capturing():
        return t.to(dt)                 # never created inside a capture: the warm-up passes before it make them
    with torch.no_grad():
        val = t.detach().to(dt)         # preserves the parameter's dense layout
    t._sd_shadow = (t._version, val)
    return val


def linear_weight_grads(x, dy2, w_shape, w_dtype, want_db, defer_ok, defer_bias_ok):
    """dW [out, in] (fp32 slabs combined at once or by the deferred pass) and, when it rides along or is cheap to take here, the bias gradient of a
    token-major Linear: x [..., in] as saved by the forward, dy2 [tokens, out] in x's dtype.  -> (dw, db or None)."""
    dw = db = None
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
    T, M, N = x2.shape[0], w_shape[0], w_shape[1]
    L = _lib.lib()
    if (_SPLIT_BF16 and _FP32_WGRAD_GROUPED and x.dtype == torch.float32 and w_dtype == torch.float32 and defer_ok and (defer_bias_ok or not want_db)
            and deferred.wgrad_groupable(dyc, x2, M, N)):
        # round 5: every fp32 weight gradient of the backward (the transposed-read split-bf16 ones, the tall-skinny exact-f32 ones, the split-K
        # and library ones of the few-token stages) joins the scope's grouped launches (one per tile width), bias sums riding along in the slabs
        return deferred.add_wgrad(dyc, x2, M, N, with_bias=want_db)
    ns_tn = 0
    if (_SPLIT_BF16 and x.dtype == torch.float32 and w_dtype == torch.float32 and dyc.data_ptr() % 16 == 0
            and x2.data_ptr() % 16 == 0):
        ns_tn = L.sd_linear_wgrad_tn_slabs(T, M, N)
    if ns_tn:
        # round 4: the tall-skinny products with out_features >= 128 (the SegFormer head over 131072 tokens) on transposed LDS reads in
        # split-bf16 arithmetic (csrc/wgrad_tn.hip) instead of the exact-f32 tall-skinny kernel; slabs combined by the deferred pass
        # the bias gradient rides along as M extra floats per slab (column sums of the staged dY values): no second pass over dY
        fuse_db = want_db and _TN_FUSED_BIAS
        slab = M * N + (M if fuse_db else 0)
        ws = torch.empty(ns_tn, slab, dtype=torch.float32, device=x.device)
        _lib.check(L.sd_linear_wgrad_tn(dyc.data_ptr(), x2.data_ptr(), ws.data_ptr(), ws.numel() * 4, T, M, N, int(fuse_db), _stream_ptr()),
                   'sd_linear_wgrad_tn')
        buf = torch.empty(slab, dtype=torch.float32, device=x.device)
        if defer_ok and deferred.enabled() and (defer_bias_ok or not fuse_db):
            deferred.add(ws, buf, slab, ns_tn)
        else:
            deferred.reduce_now(ws, buf, slab, ns_tn)
        db = buf[M * N:] if fuse_db else (deferred.column_sum(dyc, defer_bias_ok) if want_db else None)
        return buf[:M * N].view(M, N), db
    direct = bool(L.sd_linear_wgrad_fuses_bias_dtype(_DT[x.dtype], T, M, N))
    if not direct and x.dtype == torch.float32:
        # fewer than 8192 tokens or a weight of more than 16 64x64 regions: no longer tall-skinny.  Round 3: split-K over the tokens
        # on the pipelined MFMA kernel (sd_linear_wgrad_splitk), slabs combined by the deferred batched pass -- the library ran
        # these on ~100 workgroups of 32 x 32 tiles (63 us for 1024 x 256 over 2048 tokens; profiles/r03_step_shapes.txt)
        ns = L.sd_linear_wgrad_splitk_slabs(T, M, N) if _SPLITK_WGRAD and w_dtype == torch.float32 else 0
        if ns:
            ws = torch.empty(ns, M * N, dtype=torch.float32, device=x.device)
            _lib.check(L.sd_linear_wgrad_splitk(dyc.data_ptr(), x2.data_ptr(), ws.data_ptr(), ws.numel() * 4, T, M, N, _stream_ptr()),
                       'sd_linear_wgrad_splitk')
            buf = torch.empty(M * N, dtype=torch.float32, device=x.device)
            if defer_ok and deferred.enabled():
                deferred.add(ws, buf, M * N, ns)
            else:
                deferred.reduce_now(ws, buf, M * N, ns)
            dw = buf.view(M, N)
        else:
            dw = (dyc.t() @ x2).to(w_dtype)
        db = deferred.column_sum(dyc, defer_bias_ok and w_dtype == torch.float32).to(w_dtype) if want_db else None
        return dw, db
    fuse_b = want_db and direct
    if direct and defer_ok and deferred.enabled() and w_dtype == torch.float32:
        # tall-skinny plan inside a deferred scope, gradients going straight to fp32 leaf parameters: leave the split-K slabs
        # in the workspace, the scope's exit combines them together with everybody else's (segdistill_amd/deferred.py)
        slab = M * N + (M if fuse_b else 0)
        buf = torch.empty(slab, dtype=torch.float32, device=x.device)
        wsb = L.sd_linear_wgrad_workspace_bytes(T, M, N)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        deferred.side_launch(lambda: _lib.check(L.sd_linear_wgrad_partials(dyc.data_ptr(), x2.data_ptr(), _DT[x.dtype], T, M, N, int(fuse_b),
                                                                            ws.data_ptr(), wsb, _stream_ptr()), 'sd_linear_wgrad_partials'),
                             dyc, x2, ws)
        deferred.add(ws, buf, slab, L.sd_linear_wgrad_slabs(_DT[x.dtype], T, M, N))
        dw = buf[:M * N].view(M, N)
        if fuse_b:
            db = buf[M * N:]
        elif want_db:
            db = deferred.column_sum(dyc, defer_bias_ok)
        return dw, db
    if not direct and _BF16_WGRAD_LIB and x.dtype == torch.bfloat16:
        # A/B switch: the library's bf16 GEMM for the generic (not tall-skinny) weight gradients under bf16 storage.  Measured on MI355X,
        # config 5, same box: 515 imgs/s with it against 638 / 636 with the split-K kernel + deferred combine -- off by default
        dw = (dyc.t() @ x2).to(w_dtype)
        db = deferred.column_sum(dyc, defer_bias_ok and w_dtype == torch.float32).to(w_dtype) if want_db else None
        return dw, db
    if not direct and defer_ok and w_dtype == torch.float32 and x.dtype == torch.bfloat16 and deferred.wgrad_groupable(dyc, x2, M, N):
        # round 5: not even the GEMM runs now -- the scope's ONE grouped launch computes every such gradient, its k-splits planned over all of them
        if want_db and defer_bias_ok:       # the bias gradient rides along in the slabs (no second pass over dY)
            return deferred.add_wgrad(dyc, x2, M, N, with_bias=True)
        dw, _ = deferred.add_wgrad(dyc, x2, M, N)
        return dw, (deferred.column_sum(dyc, defer_bias_ok) if want_db else None)
    gs = 0 if direct else L.sd_linear_wgrad_generic_slabs(_DT[x.dtype], T, M, N)
    if gs and defer_ok and deferred.enabled() and w_dtype == torch.float32:
        # the generic split-K plan (bf16 storage: most Linears of config 5) inside a deferred scope: its slab combine joins the batched
        # pass at the end of the backward instead of running as one more launch per layer (54 of them per config-5 step)
        ws = torch.empty(gs, M * N, dtype=torch.float32, device=x.device)
        _lib.check(L.sd_linear_wgrad_generic_partials(dyc.data_ptr(), x2.data_ptr(), _DT[x.dtype], T, M, N, ws.data_ptr(), ws.numel() * 4,
                                                      _stream_ptr()), 'sd_linear_wgrad_generic_partials')
        buf = torch.empty(M * N, dtype=torch.float32, device=x.device)
        deferred.add(ws, buf, M * N, gs)
        db = deferred.column_sum(dyc, defer_bias_ok) if want_db else None
        return buf.view(M, N), db
    dw32 = torch.empty(M, N, dtype=torch.float32, device=x.device)
    db32 = torch.empty(M, dtype=torch.float32, device=x.device) if fuse_b else None
    wsb = L.sd_linear_wgrad_workspace_bytes(T, M, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    _lib.check(L.sd_linear_wgrad(dyc.data_ptr(), x2.data_ptr(), dw32.data_ptr(), None if db32 is None else db32.data_ptr(),
                                 _DT[x.dtype], T, M, N, ws.data_ptr(), wsb, _stream_ptr()), 'sd_linear_wgrad')
    dw = dw32.to(w_dtype)
    if fuse_b:
        db = db32.to(w_dtype)
    return dw, db


class _TokenLinear(torch.autograd.Function):
    """Under autocast the forward computes in the autocast dtype exactly as F.linear would (activations and a cast copy of
    the fp32 master weight in bf16); the bf16 activations are what is saved, so the split-K weight-gradient kernel reads half
    the bytes, and dW / dbias are produced in fp32 for the fp32 master parameters."""

    @staticmethod
    def forward(ctx, x, weight, bias, defer_ok=False, defer_bias_ok=False):
        ctx.defer_ok, ctx.defer_bias_ok = defer_ok, defer_bias_ok
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype('cuda')
            with torch.autocast('cuda', enabled=False):
                xc = x.to(dt)
                wc = lowp_copy(weight, dt)
                bc = None if bias is None else lowp_copy(bias, dt)
                y = linear_fwd_bf16(xc, wc, bc) if dt == torch.bfloat16 else F.linear(xc, wc, bc)
            ctx.save_for_backward(xc, wc)
        else:
            y = linear_forward(x, weight, bias)
            ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.in_dtype, ctx.w_dtype = x.dtype, weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != x.dtype:
            dy2 = dy2.to(x.dtype)
        if ctx.needs_input_grad[0]:
            dx = _bwd_data(dy2, weight)
            dx = dx.reshape(x.shape) if dx is not None else (dy2 @ weight).reshape(x.shape).to(ctx.in_dtype)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dw, db = linear_weight_grads(x, dy2, weight.shape, ctx.w_dtype, want_db, ctx.defer_ok, ctx.defer_bias_ok)
        if want_db and db is None:
            dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
            db = deferred.column_sum(dyc, ctx.defer_bias_ok and ctx.w_dtype == torch.float32).to(ctx.w_dtype)
        return dx, dw, db, None, None


def token_linear(x, weight, bias=None, defer_ok=False, defer_bias_ok=None):
    """F.linear with the HIP weight-gradient kernel when it pays (GPU, fp32/bf16 storage, many tokens, training).
    defer_ok: weight and bias are LEAF parameters whose gradients nothing reads before the optimizer, so inside a
    ``deferred.scope()`` the split-K slabs may be combined at the scope's end; defer_bias_ok: the same for the bias alone
    (default: as defer_ok) -- e.g. a weight that is a re-laid-out copy of the parameter, next to a bias that is the leaf itself."""
    amp = torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
    use = (x.is_cuda and x.dtype in _DT and (x.dtype == weight.dtype or amp) and weight.requires_grad and torch.is_grad_enabled()
           and (amp or not torch.is_autocast_enabled()) and x.numel() // x.shape[-1] >= MIN_TOKENS)
    if use:
        return _TokenLinear.apply(x, weight, bias, defer_ok, defer_ok if defer_bias_ok is None else defer_bias_ok)
    if x.is_cuda and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)):
        return linear_forward(x, weight, bias)           # frozen network (the teacher): no graph to build (autocast: cached cast copies)
    return F.linear(x, weight, bias)


def call_linear(module, x):
    """Apply an nn.Linear through token_linear unless somebody hooked the module (taps must see a module call)."""
    if module._forward_hooks or module._forward_pre_hooks:
        return module(x)
    return token_linear(x, module.weight, module.bias, defer_ok=True)


class _LongKLinear(torch.autograd.Function):
    """y = x . W^T + b for a long reduction axis and a small output (SR-attention patch projection): split-K MFMA forward;
    the two backward products are ordinary library GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = weight.contiguous()
        M, K, N = x2.shape[0], x2.shape[1], w.shape[0]
        L = _lib.lib()
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        wsb = L.sd_linear_longk_workspace_bytes(M, N, K)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        b = None if bias is None else bias.detach().float().contiguous()
        _lib.check(L.sd_linear_longk_fwd(x2.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), _DT[x2.dtype], M, N, K,
                                         ws.data_ptr(), wsb, _stream_ptr()), 'sd_linear_longk_fwd')
        ctx.save_for_backward(x2, w)
        ctx.shape, ctx.has_bias = x.shape, bias is not None
        return y.to(x.dtype).reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = (dy2 @ w).reshape(ctx.shape) if ctx.needs_input_grad[0] else None
        dw = (dy2.t() @ x2) if ctx.needs_input_grad[1] else None
        db = deferred.column_sum(dy2.contiguous(), False).to(dy2.dtype) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


class _LinearToPlanes(torch.autograd.Function):
    """y[b, :, p] = W . x[b, p, :] + bias: token-major input, contiguous class planes out (csrc: sd_linear_nchw_*).  fp32, or bf16 storage
    with fp32 master weight / bias (what autocast computes: the weight is rounded to bf16 inside the kernel)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        w = weight.detach().contiguous()
        B, P, K = x.shape
        N = w.shape[0]
        L = _lib.lib()
        y = torch.empty(B, N, P, dtype=x.dtype, device=x.device)
        b = None if bias is None else bias.detach().float().contiguous()
        if (x.dtype == torch.float32 and _SPLIT_BF16 and N <= 160 and K % 32 == 0 and planes.supported(weight) and weight.is_contiguous()
                and L.sd_get_tunable(b'align_split_bf16') == 1):
            # the weight's row-major bf16 planes, split once per optimizer step (never, for the frozen teacher): every wave of the 160-row tile
            # needs all class rows, so splitting W in registers was 5/6 of the kernel's vector work (290 -> see profiles/r03_kernels.txt)
            pr = planes.get(weight, 'rows')
            _lib.check(L.sd_linear_nchw_fwd_planes(x.data_ptr(), pr.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), _DT[x.dtype], B, P, K, N,
                                                   _stream_ptr()), 'sd_linear_nchw_fwd_planes')
        else:
            _lib.check(L.sd_linear_nchw_fwd(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), _DT[x.dtype], B, P, K, N,
                                            _stream_ptr()), 'sd_linear_nchw_fwd')
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, P, K = x.shape
        N = w.shape[0]
        L = _lib.lib()
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.check(L.sd_linear_nchw_bwd_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), _DT[x.dtype], B, P, K, N, _stream_ptr()),
                       'sd_linear_nchw_bwd_data')
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty(N, K, dtype=torch.float32, device=x.device)
            db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            wsb = L.sd_linear_nchw_workspace_bytes(B, P, K, N)
            ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
            _lib.check(L.sd_linear_nchw_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None if db is None else db.data_ptr(), _DT[x.dtype], B, P, K, N,
                                                   ws.data_ptr(), wsb, _stream_ptr()), 'sd_linear_nchw_bwd_weight')
        return dx, dw, db


def linear_to_planes_supported(x, weight, bias):
    if not (x.is_cuda and x.dim() == 3 and weight.dtype == torch.float32 and (bias is None or bias.dtype == torch.float32)
            and os.environ.get('SEGDISTILL_PRED_PLANES', '1') == '1'):
        return False
    if torch.is_autocast_enabled():
        return (x.dtype == torch.bfloat16 and torch.get_autocast_dtype('cuda') == torch.bfloat16 and x.shape[-1] % 8 == 0 and x.shape[1] % 8 == 0)
    return x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.shape[1] % 4 == 0


def linear_to_planes(x, weight, bias=None):
    """[B, P, in] tokens -> [B, out, P] planes (a 1x1 conv on a token-major map whose consumers want NCHW), with autograd."""
    return _LinearToPlanes.apply(x, weight, bias)


def patch_linear_supported(x, hw, r, weight, enabled=None):
    """The SR patch projection straight from the tokens (sd_linear_patch_fwd): fp32, no graph to build (the frozen teacher), whole patches."""
    return ((_PATCH_GEMM if enabled is None else enabled) and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous()
            and not torch.is_autocast_enabled() and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad))
            and hw[0] % r == 0 and hw[1] % r == 0 and (r * x.shape[-1]) % 16 == 0 and weight.shape[0] <= 512
            and x.shape[0] * (hw[0] // r) * (hw[1] // r) <= 16384)


def patch_linear_forward(x, hw, r, weight, bias=None):
    """y[b, (py, px), :] = W . patch(x, py, px) + bias for the r x r patches of tokens x [B, H*W, C]; W [out, r*r*C] in (ky, kx, c) order."""
    B, _, c = x.shape
    H, W = hw
    L = _lib.lib()
    rows, N, K = B * (H // r) * (W // r), weight.shape[0], r * r * c
    w = weight if weight.is_contiguous() else weight.contiguous()
    y = torch.empty(B, rows // B, N, dtype=torch.float32, device=x.device)
    wsb = L.sd_linear_longk_workspace_bytes(rows, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    b = None if bias is None else bias.detach().float().contiguous()
    _lib.check(L.sd_linear_patch_fwd(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), y.data_ptr(), B, H, W, c, r, N, ws.data_ptr(), wsb,
                                     _stream_ptr()), 'sd_linear_patch_fwd')
    return y


def longk_linear(x, weight, bias=None, weight_is_view=False):
    """F.linear for in_features >= 1024 with few rows/outputs on the GPU (fp32 weights); otherwise F.linear."""
    rows = x.numel() // x.shape[-1]
    if (_LONGK_ENABLED and x.is_cuda and x.dtype in _DT and weight.dtype == torch.float32 and x.shape[-1] >= 1024 and weight.shape[0] <= 512 and rows <= 16384
            and not torch.is_autocast_enabled()):
        return _LongKLinear.apply(x, weight, bias)
    # `weight` is a view of the conv filter (channels-last storage) or the caller's re-laid-out copy of it: the gradient of a copy is read
    # by the copy's backward at once, so only a view may be deferred; the bias is the leaf itself
    return token_linear(x, weight, bias, defer_ok=weight_is_view, defer_bias_ok=True)

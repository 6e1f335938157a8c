"""nn.Linear on token-major activations.  Forward and input gradient: the measured three-way dispatch of `_gemm_mode` below (csrc/token_gemm.hip
in split-bf16 or exact-f32 arithmetic, or the library GEMM); weight gradient dW = dY^T . X -- tall-skinny at the high-resolution MiT stages,
where the library kernels ran 30x off the HBM roofline (profiles/r01_train_step_kernels_*.txt) -- by the grouped launches of csrc/wgrad_tn.hip /
the split-K MFMA kernel (csrc/align1x1.hip: sd_linear_wgrad) with the slabs combined by the deferred batched reduction; the class-plane
`linear_pred` (sd_linear_nchw_*) lives here too.  (Rounds 1-5 carried two opt-in forms of the SR conv -- a split-K long-K forward and the patch
gather folded into the GEMM's staging -- and a library switch for the bf16 weight gradients; all three measured slower in the step
(docs/history) and were removed in round 6.)"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import _lib, deferred, planes, token_gemm
from .ops import _DT, _stream_ptr

MIN_TOKENS = 1     # every training Linear: below 8192 tokens dW is the library's GEMM, but the bias gradient still avoids ATen's
                   # memset + multi-block sum(0) pair (deferred.column_sum)


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


def _planes_ok(weight):
    """Pre-split planes are kept per PARAMETER (refreshed in place by the optimizer): a frozen weight, a trainable leaf, or a view of one.  A
    temporary that autograd made this step (a product of two parameters) would register a new entry every step."""
    return planes.supported(weight) and (not weight.requires_grad or weight.is_leaf or (weight._base is not None and weight._base.is_leaf))


def linear_forward(x, weight, bias):
    """x . W^T + bias without autograd bookkeeping (frozen networks, and the forward of _TokenLinear): the measured three-way dispatch."""
    if x.dtype == torch.float32 and token_gemm.supported(x, weight):
        mode = _gemm_mode(x.numel() // x.shape[-1], x.shape[-1], weight.shape[0])
        if mode == 'x3' and _planes_ok(weight):
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
        if mode == 'x3' and _planes_ok(weight):
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


class _WgradCall:
    """One weight-gradient product dW [M, N] = dy^T . x over T tokens, with what its caller allows (plan_weight_grad -> a launch below)."""
    __slots__ = ('x', 'dy', 'T', 'M', 'N', 'w_dtype', 'want_db', 'defer_ok', 'defer_bias_ok', 'n_tn', 'n_splitk', 'n_generic', 'direct')


def plan_weight_grad(c):
    """The kernel family that computes this product -- the MEASURED order of preference (profiles/r0N_wgrad_*bench.txt, r05_ab_cfg*_wgrad_grouped):
       'grouped'    registered with the backward's grouped launches (csrc/wgrad_tn.hip *_multi): fp32 and bf16 leaves inside a deferred scope
       'tn'         one launch of the transposed-read split-bf16 kernel (fp32, out_features >= 128 over many tokens)
       'splitk'     fp32, few tokens or a large weight: split-K on the pipelined MFMA kernel     'library'  what is left of those
       'direct'     the tall-skinny exact-f32 kernel with its slabs left to the scope's combine
       'generic'    bf16 storage outside the grouped launch: the generic split-K plan, slabs to the scope's combine
       'oneshot'    sd_linear_wgrad with its own combine (no scope, or gradients that are read at once)"""
    L = _lib.lib()
    f32 = c.x.dtype == torch.float32 and c.w_dtype == torch.float32
    if (_SPLIT_BF16 and _FP32_WGRAD_GROUPED and f32 and c.defer_ok and (c.defer_bias_ok or not c.want_db)
            and deferred.wgrad_groupable(c.dy, c.x, c.M, c.N)):
        return 'grouped'
    if _SPLIT_BF16 and f32 and c.dy.data_ptr() % 16 == 0 and c.x.data_ptr() % 16 == 0:
        c.n_tn = L.sd_linear_wgrad_tn_slabs(c.T, c.M, c.N)
        if c.n_tn:
            return 'tn'
    c.direct = bool(L.sd_linear_wgrad_fuses_bias_dtype(_DT[c.x.dtype], c.T, c.M, c.N))
    if not c.direct and c.x.dtype == torch.float32:
        c.n_splitk = L.sd_linear_wgrad_splitk_slabs(c.T, c.M, c.N) if _SPLITK_WGRAD and c.w_dtype == torch.float32 else 0
        return 'splitk' if c.n_splitk else 'library'
    if c.direct and c.defer_ok and deferred.enabled() and c.w_dtype == torch.float32:
        return 'direct'
    if (not c.direct and c.defer_ok and c.w_dtype == torch.float32 and c.x.dtype == torch.bfloat16
            and deferred.wgrad_groupable(c.dy, c.x, c.M, c.N)):
        return 'grouped'
    c.n_generic = 0 if c.direct else L.sd_linear_wgrad_generic_slabs(_DT[c.x.dtype], c.T, c.M, c.N)
    if c.n_generic and c.defer_ok and deferred.enabled() and c.w_dtype == torch.float32:
        return 'generic'
    return 'oneshot'


def _wg_grouped(c):
    # not even the GEMM runs now: the scope's grouped launches compute every such gradient, their k-splits planned over all of them; the bias
    # gradient rides along in the slabs when the caller lets it wait as well (no second pass over dY)
    if c.x.dtype == torch.float32 or (c.want_db and c.defer_bias_ok):
        return deferred.add_wgrad(c.dy, c.x, c.M, c.N, with_bias=c.want_db)
    dw, _ = deferred.add_wgrad(c.dy, c.x, c.M, c.N)
    return dw, (deferred.column_sum(c.dy, c.defer_bias_ok) if c.want_db else None)


def _wg_tn(c):
    # the tall-skinny products with out_features >= 128 (the SegFormer head over 131072 tokens) on transposed LDS reads in split-bf16
    # arithmetic (csrc/wgrad_tn.hip); the bias gradient rides along as M extra floats per slab (column sums of the staged dY values)
    L, M, N = _lib.lib(), c.M, c.N
    fuse_db = c.want_db and _TN_FUSED_BIAS
    slab = M * N + (M if fuse_db else 0)
    ws = torch.empty(c.n_tn, slab, dtype=torch.float32, device=c.x.device)
    _lib.check(L.sd_linear_wgrad_tn(c.dy.data_ptr(), c.x.data_ptr(), ws.data_ptr(), ws.numel() * 4, c.T, M, N, int(fuse_db), _stream_ptr()),
               'sd_linear_wgrad_tn')
    buf = torch.empty(slab, dtype=torch.float32, device=c.x.device)
    if c.defer_ok and deferred.enabled() and (c.defer_bias_ok or not fuse_db):
        deferred.add(ws, buf, slab, c.n_tn)
    else:
        deferred.reduce_now(ws, buf, slab, c.n_tn)
    db = buf[M * N:] if fuse_db else (deferred.column_sum(c.dy, c.defer_bias_ok) if c.want_db else None)
    return buf[:M * N].view(M, N), db


def _wg_splitk(c):
    # fewer than 8192 tokens or a weight of more than 16 64x64 regions: split-K over the tokens on the pipelined MFMA kernel, slabs combined by the
    # deferred batched pass -- the library ran these on ~100 workgroups of 32 x 32 tiles (63 us for 1024 x 256 over 2048 tokens)
    L, M, N = _lib.lib(), c.M, c.N
    ws = torch.empty(c.n_splitk, M * N, dtype=torch.float32, device=c.x.device)
    _lib.check(L.sd_linear_wgrad_splitk(c.dy.data_ptr(), c.x.data_ptr(), ws.data_ptr(), ws.numel() * 4, c.T, M, N, _stream_ptr()), 'sd_linear_wgrad_splitk')
    buf = torch.empty(M * N, dtype=torch.float32, device=c.x.device)
    if c.defer_ok and deferred.enabled():
        deferred.add(ws, buf, M * N, c.n_splitk)
    else:
        deferred.reduce_now(ws, buf, M * N, c.n_splitk)
    db = deferred.column_sum(c.dy, c.defer_bias_ok and c.w_dtype == torch.float32).to(c.w_dtype) if c.want_db else None
    return buf.view(M, N), db


def _wg_library(c):
    dw = (c.dy.t() @ c.x).to(c.w_dtype)
    db = deferred.column_sum(c.dy, c.defer_bias_ok and c.w_dtype == torch.float32).to(c.w_dtype) if c.want_db else None
    return dw, db


def _wg_direct(c):
    # tall-skinny plan inside a deferred scope, gradients going straight to fp32 leaf parameters: the split-K slabs stay in the workspace,
    # the scope's exit combines them together with everybody else's (segdistill_amd/deferred.py)
    L, M, N, x2, dyc = _lib.lib(), c.M, c.N, c.x, c.dy
    fuse_b = c.want_db
    slab = M * N + (M if fuse_b else 0)
    buf = torch.empty(slab, dtype=torch.float32, device=x2.device)
    wsb = L.sd_linear_wgrad_workspace_bytes(c.T, M, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x2.device)
    deferred.side_launch(lambda: _lib.check(L.sd_linear_wgrad_partials(dyc.data_ptr(), x2.data_ptr(), _DT[x2.dtype], c.T, M, N, int(fuse_b),
                                                                        ws.data_ptr(), wsb, _stream_ptr()), 'sd_linear_wgrad_partials'),
                         dyc, x2, ws)
    deferred.add(ws, buf, slab, L.sd_linear_wgrad_slabs(_DT[x2.dtype], c.T, M, N))
    return buf[:M * N].view(M, N), (buf[M * N:] if fuse_b else None)


def _wg_generic(c):
    # the generic split-K plan (bf16 storage) inside a deferred scope: its slab combine joins the batched pass at the end of the backward
    L, M, N = _lib.lib(), c.M, c.N
    ws = torch.empty(c.n_generic, M * N, dtype=torch.float32, device=c.x.device)
    _lib.check(L.sd_linear_wgrad_generic_partials(c.dy.data_ptr(), c.x.data_ptr(), _DT[c.x.dtype], c.T, M, N, ws.data_ptr(), ws.numel() * 4,
                                                  _stream_ptr()), 'sd_linear_wgrad_generic_partials')
    buf = torch.empty(M * N, dtype=torch.float32, device=c.x.device)
    deferred.add(ws, buf, M * N, c.n_generic)
    return buf.view(M, N), (deferred.column_sum(c.dy, c.defer_bias_ok) if c.want_db else None)


def _wg_oneshot(c):
    L, M, N = _lib.lib(), c.M, c.N
    fuse_b = c.want_db and c.direct
    dw32 = torch.empty(M, N, dtype=torch.float32, device=c.x.device)
    db32 = torch.empty(M, dtype=torch.float32, device=c.x.device) if fuse_b else None
    wsb = L.sd_linear_wgrad_workspace_bytes(c.T, M, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=c.x.device)
    _lib.check(L.sd_linear_wgrad(c.dy.data_ptr(), c.x.data_ptr(), dw32.data_ptr(), None if db32 is None else db32.data_ptr(),
                                 _DT[c.x.dtype], c.T, M, N, ws.data_ptr(), wsb, _stream_ptr()), 'sd_linear_wgrad')
    return dw32.to(c.w_dtype), (db32.to(c.w_dtype) if fuse_b else None)


_WGRAD_LAUNCH = {'grouped': _wg_grouped, 'tn': _wg_tn, 'splitk': _wg_splitk, 'library': _wg_library, 'direct': _wg_direct, 'generic': _wg_generic,
                 'oneshot': _wg_oneshot}


def linear_weight_grads(x, dy2, w_shape, w_dtype, want_db, defer_ok, defer_bias_ok):
    """dW [out, in] (fp32 slabs combined at once or by the deferred pass) and, when it rides along or is cheap to take here, the bias gradient of a
    token-major Linear: x [..., in] as saved by the forward, dy2 [tokens, out] in x's dtype.  -> (dw, db or None: the caller then sums dy2's
    columns itself).  Two steps: plan_weight_grad picks the kernel family, _WGRAD_LAUNCH[plan] runs it."""
    c = _WgradCall()
    x2 = x.reshape(-1, x.shape[-1])
    c.x = x2 if x2.is_contiguous() else x2.contiguous()
    c.dy = dy2 if dy2.is_contiguous() else dy2.contiguous()
    c.T, c.M, c.N = c.x.shape[0], w_shape[0], w_shape[1]
    c.w_dtype, c.want_db, c.defer_ok, c.defer_bias_ok = w_dtype, want_db, defer_ok, defer_bias_ok
    c.n_tn = c.n_splitk = c.n_generic = 0
    c.direct = False
    return _WGRAD_LAUNCH[plan_weight_grad(c)](c)


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


def sr_patch_linear(patches, weight, bias=None, weight_is_view=False):
    """The spatial-reduction conv of the MiT attention (kernel == stride == r: a Linear over r x r patches, mix_transformer.py:86-88,112-116) on
    the gathered patch matrix.  `weight` is a view of the conv filter (channels-last storage) or the caller's re-laid-out copy of it: the gradient
    of a copy is read by the copy's backward at once, so only a view may be deferred; the bias is the leaf itself."""
    return token_linear(patches, weight, bias, defer_ok=weight_is_view, defer_bias_ok=True)

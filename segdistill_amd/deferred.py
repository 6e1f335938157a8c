"""Deferred combination of parameter-gradient partials (csrc/reduce.hip).

The LayerNorm and Linear weight-gradient kernels produce per-workgroup partial slabs and used to combine them with one small
launch each (~65 per backward of Segformer-B0).  Inside ``scope()`` they leave the slabs in their workspace and register a job
here; the scope's exit combines ALL jobs in one launch per 80 (``sd_multi_slab_reduce``).  Valid because nothing reads those
gradients before the optimizer: a call site opts in only when its result goes straight to a leaf parameter's ``.grad`` (no cast,
no stacking, no accumulation into an existing gradient).  Outside a scope every op combines its partials at once, as before.

Round 5: the weight-gradient GEMMs themselves are deferred too.  ``add_wgrad`` / ``add_dw_wgrad`` only REGISTER a Linear's / a depth-wise
convolution's filter gradient (operands held alive, results handed out as views of buffers that are written at the flush); the scope's exit
plans the k-splits of all of them together and computes them in grouped launches (csrc/wgrad_tn.hip ``sd_linear_wgrad_tn_multi``, csrc/dwconv.hip
``sd_dwconv3x3_wgrad_multi``), then runs the one slab combine.  DESIGN 3.5b."""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch

from . import _lib
from .ops import _DT, _stream_ptr

_jobs = None   # None: not deferring.  list of (partials tensor, out tensor, n, nslabs) -- the tensors are held until the flush
_colsums = []  # column sums whose PARTIALS are deferred as well: (x2d, partials, out, rows, C, nblk), x2d kept alive until the flush
_wgrads = []   # bf16 weight gradients whose GEMM is deferred as well: (dY, X, out, tokens, M, N), operands kept alive until the flush
_dw_wgrads = []  # depth-wise 3x3 filter gradients whose partials launch is deferred: (x, dy, ws, B, H, W, C), grouped into one launch at the flush
_side = {}     # device index -> the side stream the weight-gradient kernels of a scope run on
_held = []     # operands of side-stream launches, kept alive until the join (so the allocator cannot hand their memory out earlier)
_forked = None  # the side stream with un-joined work, if any
_held_ptrs = set()   # storages of the operands a scope keeps alive for its deferred launches (dY / X of every registered weight gradient)
_held_bytes = 0
partial_flushes = 0  # diagnostic: how many times a scope ran its grouped launches early because the budget below was reached

# deferring the weight-gradient GEMMs keeps every Linear's dY and saved X (and every depth-wise convolution's x / dy) alive until the
# scope ends, where rounds 2-4 freed them as the backward progressed.  The budget bounds that: once the registered operands exceed it, the
# grouped launches registered SO FAR run at once (their results are only read after the scope's combine, so an early launch is always valid)
# and the operands are released.  Default: 1/8 of the device memory (36 GB on MI355X; config 2 holds 1.9 GB, config 5 3.3 GB: never reached
# by the BASELINE configs).  SEGDISTILL_WGRAD_HELD_MB overrides it (0: flush at every registration = the pre-round-5 memory profile).
_HELD_BUDGET_MB = os.environ.get('SEGDISTILL_WGRAD_HELD_MB')


def held_budget_bytes(device=None):
    if _HELD_BUDGET_MB is not None:
        return int(float(_HELD_BUDGET_MB) * (1 << 20))
    try:
        return torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).total_memory // 8
    except Exception:  # noqa: BLE001 -- no GPU (CPU tests of the bookkeeping)
        return 1 << 62


def held_bytes():
    return _held_bytes


def _note_held(*tensors):
    """Account the operands a registration keeps alive; when the scope's budget is reached, run the grouped launches registered so far."""
    global _held_bytes
    for t in tensors:
        if t is None:
            continue
        try:
            key = t.untyped_storage().data_ptr()
            nbytes = t.untyped_storage().nbytes()
        except Exception:  # noqa: BLE001
            key, nbytes = t.data_ptr(), t.numel() * t.element_size()
        if key not in _held_ptrs:
            _held_ptrs.add(key)
            _held_bytes += nbytes
    if _held_bytes > held_budget_bytes(tensors[0].device if tensors and tensors[0].is_cuda else None):
        global partial_flushes
        partial_flushes += 1
        flush_operands()


def flush_operands():
    """Run every deferred launch that holds operands (weight-gradient GEMMs, depth-wise filter partials, column-sum partials) NOW; their slab
    combines stay with the scope's exit.  Safe at any point inside a scope: nobody reads these gradients before the scope ends."""
    global _held_bytes
    if _wgrads:
        _flush_wgrads()
    if _dw_wgrads:
        _flush_dw_wgrads()
    if _colsums:
        _flush_colsums()
    _held_ptrs.clear()
    _held_bytes = 0


class _ColsumJob(C.Structure):
    _fields_ = [('x', C.c_void_p), ('partials', C.c_void_p), ('rows', C.c_long), ('C', C.c_int), ('reserved', C.c_int)]


class _WgradJob(C.Structure):
    _fields_ = [('dY', C.c_void_p), ('X', C.c_void_p), ('slabs', C.c_void_p), ('tokens', C.c_long), ('out_features', C.c_int), ('in_features', C.c_int),
                ('nsplit', C.c_int), ('with_bias', C.c_int)]


class _DwWgradJob(C.Structure):
    _fields_ = [('x', C.c_void_p), ('dy', C.c_void_p), ('partials', C.c_void_p), ('partials_bytes', C.c_size_t), ('B', C.c_int), ('H', C.c_int),
                ('W', C.c_int), ('C', C.c_int)]


class _Job(C.Structure):
    _fields_ = [('partials', C.c_void_p), ('out', C.c_void_p), ('n', C.c_long), ('nslabs', C.c_int), ('reserved', C.c_int)]


def enabled():
    return _jobs is not None


def add(partials, out, n, nslabs):
    """out[i] = sum_s partials[s*n + i], i < n -- to be computed when the enclosing scope ends.  Both tensors fp32 on the GPU.
    The caller must hand autograd VIEWS of `out`, never `out` itself (see column_sum)."""
    _jobs.append((partials, out, int(n), int(nslabs)))


# SEGDISTILL_WGRAD_GROUPED=0 (A/B): every weight gradient as its own launch inside the backward, planned on its own (rounds 2-4)
_WGRAD_GROUPED = os.environ.get('SEGDISTILL_WGRAD_GROUPED', '1') == '1'


def wgrad_groupable(dy2, x2, M, N):
    """May this weight gradient dY^T . X join the scope's grouped launch (csrc/wgrad_tn.hip: wgrad_tn_bf16_ring_multi / wgrad_tn_x3_multi)?"""
    return (_jobs is not None and _WGRAD_GROUPED and dy2.dtype == x2.dtype and dy2.dtype in _DT and dy2.is_contiguous()
            and x2.is_contiguous() and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0
            and bool(_lib.lib().sd_linear_wgrad_tn_multi_supported(_DT[dy2.dtype], dy2.shape[0], M, N)))


def add_wgrad(dy2, x2, M, N, with_bias=False):
    """dW [M, N] fp32 = dy2 [T, M]^T . x2 [T, N], computed when the enclosing scope ends by the grouped launch -- its k-splits planned over all the
    scope's weight gradients together -- and the scope's slab combine.  with_bias: the column sums of dy2 (the Linear's bias gradient) ride along in
    the slabs.  Returns VIEWS of the result buffer (see column_sum): (dW, db or None)."""
    with_bias = bool(with_bias)
    out = torch.empty(M * N + (M if with_bias else 0), dtype=torch.float32, device=dy2.device)
    _wgrads.append((dy2, x2, out, int(dy2.shape[0]), int(M), int(N), with_bias))
    views = out[:M * N].view(M, N), (out[M * N:] if with_bias else None)
    _note_held(dy2, x2)
    return views


def _flush_wgrads():
    global _wgrads
    pend_all, _wgrads = _wgrads, []
    L = _lib.lib()
    for dt in (torch.bfloat16, torch.float32):      # one plan + one launch group per storage type
        pend = [p for p in pend_all if p[0].dtype == dt]
        if not pend:
            continue
        arr = (_WgradJob * len(pend))()
        for k, (dy, x, out, T, M, N, wb) in enumerate(pend):
            arr[k].dY, arr[k].X, arr[k].tokens, arr[k].out_features, arr[k].in_features, arr[k].with_bias = dy.data_ptr(), x.data_ptr(), T, M, N, int(wb)
        _lib.check(L.sd_linear_wgrad_tn_multi_plan(C.cast(arr, C.c_void_p), len(pend), _DT[dt]), 'sd_linear_wgrad_tn_multi_plan')
        # every slab is rounded up to a multiple of 4 floats: 16-byte aligned slab origins whatever M is
        sizes = [((M * N + (M if wb else 0)) + 3) // 4 * 4 for (_, _, _, _, M, N, wb) in pend]
        total = sum(arr[k].nsplit * sizes[k] for k in range(len(pend)) if arr[k].nsplit > 1)
        ws = torch.empty(max(total, 4), dtype=torch.float32, device=pend[0][0].device)
        off = 0
        for k, (dy, x, out, T, M, N, wb) in enumerate(pend):
            ns, n = arr[k].nsplit, M * N + (M if wb else 0)
            if ns > 1:
                assert n == sizes[k], 'slab size must be a multiple of 4 floats (out % 8 == 0)'
                part = ws[off:off + ns * n]
                off += ns * n
                arr[k].slabs = part.data_ptr()
                _jobs.append((part, out, n, ns))
            else:
                arr[k].slabs = out.data_ptr()          # one split: the product IS the gradient
        _lib.check(L.sd_linear_wgrad_tn_multi(C.cast(arr, C.c_void_p), len(pend), _DT[dt], _stream_ptr()), 'sd_linear_wgrad_tn_multi')
        # (the slab views in _jobs keep `ws` alive until the combine is enqueued; the operands in `pend_all` until here)


def add_dw_wgrad(x, dy, ws, B, H, W, Cc):
    """The partials launch of a depth-wise 3 x 3 filter gradient (csrc/dwconv.hip), deferred: all of a scope's run as ONE launch when it ends
    (the caller registers the combine of `ws` with add() as before)."""
    _dw_wgrads.append((x, dy, ws, int(B), int(H), int(W), int(Cc)))
    _note_held(x, dy)


def _flush_dw_wgrads():
    global _dw_wgrads
    pend, _dw_wgrads = _dw_wgrads, []
    for dt in {p[0].dtype for p in pend}:
        group = [p for p in pend if p[0].dtype == dt]
        arr = (_DwWgradJob * len(group))()
        for k, (x, dy, ws, B, H, W, Cc) in enumerate(group):
            arr[k].x, arr[k].dy, arr[k].partials, arr[k].partials_bytes = x.data_ptr(), dy.data_ptr(), ws.data_ptr(), ws.numel() * ws.element_size()
            arr[k].B, arr[k].H, arr[k].W, arr[k].C = B, H, W, Cc
        _lib.check(_lib.lib().sd_dwconv3x3_wgrad_multi(C.cast(arr, C.c_void_p), len(group), _DT[dt], _stream_ptr()), 'sd_dwconv3x3_wgrad_multi')


def reduce_now(partials, out, n, nslabs):
    """out[i] = sum_s partials[s*n + i] at once (one launch of the batched kernel with a single job)."""
    job = (_Job * 1)()
    job[0].partials, job[0].out, job[0].n, job[0].nslabs = partials.data_ptr(), out.data_ptr(), int(n), int(nslabs)
    _lib.check(_lib.lib().sd_multi_slab_reduce(C.cast(job, C.c_void_p), 1, _stream_ptr()), 'sd_multi_slab_reduce')


def side_launch(fn, *operands):
    """Inside a scope: run `fn` (kernel launches whose results nobody reads before the scope ends -- weight-gradient partials) on a SIDE
    stream, ordered behind everything enqueued so far on the current stream, so that they overlap the backward's critical chain
    (dX GEMM -> LayerNorm -> attention ...) instead of sitting in it.  Outside a scope `fn` runs inline.
    Memory safety: every operand tensor is allocated on the main stream and a reference is held until join(), which makes the main
    stream wait for the side stream; nothing is freed -- hence nothing reused -- while side-stream work may still touch it.
    OPT-IN (SEGDISTILL_WGRAD_STREAM=1).  A/B on MI355X, config 2, replayed graph: 558 imgs/s with the fork against 608 without -- the
    ~35 fork/join pairs become cross-stream dependencies inside the hipGraph and the weight-gradient kernels take CUs from the chain
    they were meant to hide behind; the teacher stream already fills the gaps that exist.  Kept for bisecting, off by default."""
    global _forked
    if _jobs is None or os.environ.get('SEGDISTILL_WGRAD_STREAM', '0') != '1':
        fn()
        return
    main = torch.cuda.current_stream()
    side = _side.get(main.device_index)
    if side is None:
        side = _side[main.device_index] = torch.cuda.Stream(device=main.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        fn()
    _held.extend(operands)
    _forked = side


def join():
    """Make the current stream wait for the side stream's work of this scope (called before the combines, and by the segmented
    graph recorder before it ends a graph: a capture may only end with every forked stream joined)."""
    global _forked
    if _forked is not None:
        torch.cuda.current_stream().wait_stream(_forked)
        _forked = None
    _held.clear()


def _flush_colsums():
    global _colsums
    pend, _colsums = _colsums, []
    for dt in {p[0].dtype for p in pend}:          # one batched partials launch per storage type
        group = [p for p in pend if p[0].dtype == dt]
        arr = (_ColsumJob * len(group))()
        for k, (x, part, out, rows, Cc, nblk) in enumerate(group):
            arr[k].x, arr[k].partials, arr[k].rows, arr[k].C = x.data_ptr(), part.data_ptr(), rows, Cc
        _lib.check(_lib.lib().sd_multi_colsum_partials(C.cast(arr, C.c_void_p), len(group), _DT[dt], _stream_ptr()), 'sd_multi_colsum_partials')
    for (x, part, out, rows, Cc, nblk) in pend:
        _jobs.append((part, out, Cc, nblk))


def flush():
    global _jobs
    join()
    flush_operands()
    if not _jobs:
        return
    jobs, _jobs = _jobs, []
    arr = (_Job * len(jobs))()
    for k, (part, out, n, ns) in enumerate(jobs):
        arr[k].partials, arr[k].out, arr[k].n, arr[k].nslabs = part.data_ptr(), out.data_ptr(), n, ns
    _lib.check(_lib.lib().sd_multi_slab_reduce(C.cast(arr, C.c_void_p), len(jobs), _stream_ptr()), 'sd_multi_slab_reduce')


@contextlib.contextmanager
def scope():
    """Defer the combines of everything that runs inside (typically ``loss.backward()``); they are issued when the scope ends,
    on the current stream.  Nested scopes join the outer one."""
    global _jobs
    if _jobs is not None:
        yield
        return
    _jobs = []
    try:
        yield
        flush()
    finally:
        join()
        _jobs = None
        _colsums.clear()
        _wgrads.clear()
        _dw_wgrads.clear()
        _held_ptrs.clear()
        globals()['_held_bytes'] = 0


def column_sum(x2d, defer_ok=True):
    """sum over the rows of a token-major matrix [rows, C] -> [C] fp32 (a Linear's bias gradient): one partials launch, combined by the
    enclosing scope's batched pass (when the caller says nothing reads the result before the scope ends) -- or at once.  Falls back
    to ATen for layouts the kernel does not take."""
    rows, Cc = x2d.shape
    if not (x2d.is_cuda and x2d.dtype in _DT and x2d.is_contiguous() and Cc % 4 == 0 and Cc <= 8192 and rows > 0):
        return x2d.sum(0, dtype=torch.float32)
    L = _lib.lib()
    nblk = L.sd_colsum_blocks(rows, Cc)
    part = torch.empty(nblk, Cc, dtype=torch.float32, device=x2d.device)
    out = torch.empty(Cc, dtype=torch.float32, device=x2d.device)
    if enabled() and defer_ok:
        # nothing at all is launched now: the matrix is kept alive and its columns are summed, together with everybody else's, when the
        # scope ends (one partials launch + the shared combine) -- the per-layer launch leaves the backward's critical chain
        _colsums.append((x2d, part, out, rows, Cc, nblk))
        _note_held(x2d)
        # hand out a VIEW: autograd's AccumulateGrad keeps ("steals") a gradient tensor only if nobody else references that tensor
        # object, and clones it otherwise -- a clone taken before the flush would freeze the not-yet-written values.  The job list holds
        # `out`; a view is its own tensor object on the same storage.  (Every deferring op returns views of its job's buffer.)
        return out.view(Cc)
    _lib.check(L.sd_colsum_partials(x2d.data_ptr(), _DT[x2d.dtype], rows, Cc, part.data_ptr(), part.numel() * 4, _stream_ptr()), 'sd_colsum_partials')
    job = (_Job * 1)()
    job[0].partials, job[0].out, job[0].n, job[0].nslabs = part.data_ptr(), out.data_ptr(), Cc, nblk
    _lib.check(L.sd_multi_slab_reduce(C.cast(job, C.c_void_p), 1, _stream_ptr()), 'sd_multi_slab_reduce')
    return out

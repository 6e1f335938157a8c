"""Deferred combination of parameter-gradient partials (csrc/reduce.hip).

The LayerNorm and Linear weight-gradient kernels produce per-workgroup partial slabs and used to combine them with one small
launch each (~65 per backward of Segformer-B0).  Inside ``scope()`` they leave the slabs in their workspace and register a job
here; the scope's exit combines ALL jobs in one launch per 24 (``sd_multi_slab_reduce``).  Valid because nothing reads those
gradients before the optimizer: a call site opts in only when its result goes straight to a leaf parameter's ``.grad`` (no cast,
no stacking, no accumulation into an existing gradient).  Outside a scope every op combines its partials at once, as before."""
from __future__ import annotations

import contextlib
import ctypes as C

from . import _lib
from .ops import _stream_ptr

_jobs = None   # None: not deferring.  list of (partials tensor, out tensor, n, nslabs) -- the tensors are held until the flush


class _Job(C.Structure):
    _fields_ = [('partials', C.c_void_p), ('out', C.c_void_p), ('n', C.c_long), ('nslabs', C.c_int), ('reserved', C.c_int)]


def enabled():
    return _jobs is not None


def add(partials, out, n, nslabs):
    """out[i] = sum_s partials[s*n + i], i < n -- to be computed when the enclosing scope ends.  Both tensors fp32 on the GPU."""
    _jobs.append((partials, out, int(n), int(nslabs)))


def flush():
    global _jobs
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
        _jobs = None

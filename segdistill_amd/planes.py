"""Pre-split weight planes for the split-bf16 Linear products (csrc/token_gemm.hip: presplit_planes, sd_linear_*_planes).

A weight changes once per optimizer step (the frozen teacher's never), so its three bf16 planes -- in the fragment-contiguous order the
matrix cores consume -- are written once per change instead of being re-derived from fp32 inside every k-step of every GEMM that reads it:
  * frozen weights: computed at first use, cached on the tensor (in-place version counter + data pointer, like layers.frozen_derived);
  * trainable weights: an `Entry` per distinct weight VIEW (a Linear's whole matrix, or a column block of one -- the per-branch blocks of
    the SegFormer head's linear_fuse), computed at first use and from then on REFRESHED IN PLACE by the optimizer: engine/optim.py::HipAdamW
    writes parameters through raw pointers (no version bump) and calls `refresh(updated parameters)` right after its launch -- ONE
    sd_presplit_multi launch for all of them.  The buffers never move, so a captured hipGraph keeps reading the right addresses.
Freshness has ONE rule, shared by the forward and the optimizer (two predicates drifted apart for the bf16 shadows): an entry is
fresh iff `entry.version == weight._version` and its base parameter is alive.  Anything that writes the parameter through torch (checkpoint
load, another optimizer, `copy_`) bumps the version and the next forward recomputes in place -- also inside a capture, where the recompute
simply becomes a node of the graph.  NOT every torch write bumps the version: `param.data` writes and torch's FUSED optimizers
(`fused=True`: `_fused_adamw_` / `_fused_sgd_`) leave it unchanged -- call `invalidate(param)` / `sync(params)` after such a write;
engine/optim.py::build_optimizer registers a step post-hook that does so for every torch optimizer it builds.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib
from .ops import _stream_ptr


class _Job(C.Structure):
    _fields_ = [('W', C.c_void_p), ('w_row_stride', C.c_long), ('out_features', C.c_int), ('in_features', C.c_int), ('fwd_planes', C.c_void_p),
                ('bwd_planes', C.c_void_p), ('row_planes', C.c_void_p)]


class Entry:
    __slots__ = ('base', 'key', 'w_ptr', 'ldw', 'out', 'inp', 'fwd', 'bwd', 'rows', 'version', '__weakref__')

    def fill(self, job):
        job.W, job.w_row_stride, job.out_features, job.in_features = self.w_ptr, self.ldw, self.out, self.inp
        job.fwd_planes = None if self.fwd is None else self.fwd.data_ptr()
        job.bwd_planes = None if self.bwd is None else self.bwd.data_ptr()
        job.row_planes = None if self.rows is None else self.rows.data_ptr()


_ENTRIES = {}      # key -> Entry (trainable weights)
_BY_BASE = {}      # id(base parameter) -> [Entry]
_GEN = 0           # bumped whenever the set of entries or of their buffers changes (the optimizer caches its job table against it)


def supported(weight):
    return (weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and weight.stride(1) == 1
            and weight.stride(0) >= weight.shape[1])


def _launch(entries):
    arr = (_Job * len(entries))()
    for j, e in zip(arr, entries):
        e.fill(j)
    _lib.check(_lib.lib().sd_presplit_multi(C.cast(arr, C.c_void_p), len(entries), _stream_ptr()), 'sd_presplit_multi')


def _alloc_rows(out, inp, device):
    nbytes = _lib.lib().sd_presplit_rows_bytes(int(out), int(inp))
    if nbytes == 0:
        raise ValueError(f'row planes need in_features % 8 == 0 (got {inp})')
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _alloc(n_cols, k_depth, device):
    nbytes = _lib.lib().sd_presplit_bytes(int(n_cols), int(k_depth))
    if nbytes == 0:
        raise ValueError(f'pre-split planes need a reduction depth that is a multiple of 16 (got {k_depth})')
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _drop(key, base_id):
    global _GEN
    if _ENTRIES is None:            # interpreter shutdown
        return
    e = _ENTRIES.pop(key, None)
    if e is not None:
        lst = _BY_BASE.get(base_id)
        if lst is not None:
            lst[:] = [x for x in lst if x is not e]
            if not lst:
                _BY_BASE.pop(base_id, None)
        _GEN += 1


def get(weight, direction):
    """-> the uint8 planes tensor of `weight` [out, in] for direction 'fwd' (y = x . W^T, reduction over in), 'bwd' (dx = dy . W, reduction
    over out) -- both in fragment order -- or 'rows' (row-major planes [3][out][in]: the weight as the LDS-staged A operand of
    sd_linear_nchw_fwd_planes), fresh for the weight's current values."""
    global _GEN
    out, inp = weight.shape
    if not weight.requires_grad:
        from .layers import frozen_derived

        def make():
            e = Entry()
            e.w_ptr, e.ldw, e.out, e.inp = weight.data_ptr(), weight.stride(0), out, inp
            e.fwd = _alloc(out, inp, weight.device) if direction == 'fwd' else None
            e.bwd = _alloc(inp, out, weight.device) if direction == 'bwd' else None
            e.rows = _alloc_rows(out, inp, weight.device) if direction == 'rows' else None
            _launch([e])
            return getattr(e, direction)
        root = weight._base if weight._base is not None else weight        # a view object may be new on every call: cache on its base
        return frozen_derived(root, ('planes', direction, weight.data_ptr(), out, inp, weight.stride(0)), make)
    base = weight._base if weight._base is not None else weight
    key = (id(base), weight.data_ptr(), out, inp, weight.stride(0))
    e = _ENTRIES.get(key)
    if e is not None and e.base() is not base:          # a dead parameter's id was reused
        _drop(key, id(base))
        e = None
    if e is None:
        e = Entry()
        e.base = weakref.ref(base, lambda _, k=key, b=id(base), drop=_drop: drop(k, b))     # `drop` bound now: module globals are gone at interpreter exit
        e.key, e.w_ptr, e.ldw, e.out, e.inp, e.fwd, e.bwd, e.rows, e.version = key, weight.data_ptr(), weight.stride(0), out, inp, None, None, None, None
        _ENTRIES[key] = e
        _BY_BASE.setdefault(id(base), []).append(e)
        _GEN += 1
    want = getattr(e, direction)
    if want is None:
        if direction == 'rows':
            e.rows = _alloc_rows(out, inp, weight.device)
        else:
            # both fragment-order directions of a trainable weight are written together (one launch now, one per step from the optimizer)
            if inp % 16 == 0 and e.fwd is None:
                e.fwd = _alloc(out, inp, weight.device)
            if out % 16 == 0 and e.bwd is None:
                e.bwd = _alloc(inp, out, weight.device)
        e.version = None
        _GEN += 1
        want = getattr(e, direction)
        if want is None:
            raise ValueError(f'no {direction} planes for a weight of shape {tuple(weight.shape)}')
    if e.version != weight._version:
        _launch([e])
        e.version = weight._version
    return want


def invalidate(param=None):
    """Forget what is known about `param` (all parameters when None): the next forward recomputes its planes in place."""
    for e in (list(_ENTRIES.values()) if param is None else list(_BY_BASE.get(id(param), []))):
        e.version = None


def sync(params=None):
    """Recompute NOW, in place, the planes of `params` (all when None) and mark them fresh -- for code that has just changed weights behind a
    captured graph's back (KDTrainer.load_state_dict after enable_graph: a replay runs no Python forward that could notice the stale version)."""
    ents = list(_ENTRIES.values()) if params is None else [e for p in params for e in _BY_BASE.get(id(p), ())]
    ents = [e for e in ents if e.base() is not None and (e.fwd is not None or e.bwd is not None or e.rows is not None)]
    if ents:
        _launch(ents)
        for e in ents:
            e.version = e.base()._version
    return len(ents)


class Refresher:
    """The optimizer's side: refresh(params) rewrites, in ONE launch, the planes of every entry whose base parameter is in `params`
    (the tensors the optimizer has just updated through raw pointers).  The ctypes job table is cached against the registry generation
    and the identity of the parameter list."""

    def __init__(self):
        self._gen, self._ids, self._arr, self._n = -1, None, None, 0

    def refresh(self, params):
        if not _ENTRIES:
            return 0
        ids = tuple(id(p) for p in params)
        if self._gen != _GEN or self._ids != ids:
            ents = [e for pid in ids for e in _BY_BASE.get(pid, ()) if e.base() is not None and (e.fwd is not None or e.bwd is not None or e.rows is not None)]
            # entries that are STALE by the shared rule stay stale (their next forward recomputes them); fresh ones are kept fresh
            self._ents = ents
            self._arr = (_Job * max(1, len(ents)))()
            for j, e in zip(self._arr, ents):
                e.fill(j)
            self._n, self._gen, self._ids = len(ents), _GEN, ids
        if self._n:
            _lib.check(_lib.lib().sd_presplit_multi(C.cast(self._arr, C.c_void_p), self._n, _stream_ptr()), 'sd_presplit_multi')
        return self._n

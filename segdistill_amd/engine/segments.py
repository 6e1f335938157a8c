"""Segmented hipGraph capture: ONE training step recorded as a chain of graphs with the collectives BETWEEN them.

Why.  With more than one rank the student's SyncBatchNorm exchanges statistics in the middle of the forward (an
all-gather) and of the backward (an all-reduce) -- reference segformer_head.py:66-71 builds ``linear_fuse`` with SyncBN
whatever the config says.  Recording RCCL collectives into a hipGraph is avoided on purpose (engine/trainer.py), so the
whole-step capture of the single-rank path used to degrade to the `hybrid` mode (backbone graphs + eager head / losses:
~12 % slower, host-bound).  Here the step is still captured whole, but the capture is CUT at every collective:

    graph 0 | all_gather(stats) | graph 1 | all_reduce(dy sums) | graph 2

At replay the graphs are launched in order with the collectives issued eagerly between them on the same stream, on
buffers whose addresses are static.  All graphs share one memory pool and are always replayed in capture order, which
is the condition under which torch lets sequential captures share a pool.

Cutting the BACKWARD on the host thread needs the autograd graph cut as well (the engine runs device nodes on its own
worker thread, from which a capture cannot be ended): ``ChainedSyncBatchNorm`` (layers.py) normalises a DETACHED copy of
its input and leaves a record here; after ``loss.backward()`` the trainer calls ``finish_backward()``, which walks the
records in reverse creation order: reduce dy -> CUT -> all-reduce -> input gradient -> ``x.backward(dx)`` into the
upstream part of the autograd graph.  The arithmetic is torch.nn.SyncBatchNorm's own ATen sequence
(batch_norm_stats / gather_stats_with_counts / elemt / backward_reduce / backward_elemt), so a chained step and a plain
torch SyncBatchNorm step agree to rounding.
"""
from __future__ import annotations

import gc

import torch
import torch.distributed as dist

MAX_CHAINED_NORMS = 8   # more cuts than this (ResNet students: 28 SyncBN layers) and the hybrid mode is the better trade


class _Record:
    __slots__ = ('mod', 'x', 'xd', 'mean', 'invstd', 'counts', 'dy', 'red')

    def __init__(self, mod, x, xd, mean, invstd, counts):
        self.mod, self.x, self.xd, self.mean, self.invstd, self.counts = mod, x, xd, mean, invstd, counts
        self.dy = None
        self.red = None


class _ChainedNormApply(torch.autograd.Function):
    """y = (xd - mean) * invstd * w + b with the statistics as constants; the backward only COLLECTS dy (the statistics'
    share of the input gradient needs an all-reduce first and is added by SegmentRecorder.finish_backward)."""

    @staticmethod
    def forward(ctx, xd, weight, bias, mean, invstd, eps, rec):
        ctx.rec = rec
        return torch.batch_norm_elemt(xd, weight, bias, mean, invstd, eps)

    @staticmethod
    def backward(ctx, dy):
        rec = ctx.rec
        rec.dy = dy if rec.dy is None else rec.dy + dy
        return None, None, None, None, None, None, None


def _channels_first_dense(t):
    if t.is_contiguous(memory_format=torch.channels_last) or t.is_contiguous():
        return t
    return t.contiguous()


class SegmentRecorder:
    def __init__(self):
        self.items = []          # CUDAGraph | callable (a collective), in replay order
        self.records = []        # chained norms of the step being run, in forward order
        self.capturing = False
        self._graph = None
        self._pool = None
        self._stream_ctx = None
        self.cuts = 0

    # ---- capture / replay ---------------------------------------------------------------------------------------------
    def _begin_graph(self):
        self._graph = torch.cuda.CUDAGraph()
        self._graph.capture_begin(pool=self._pool)

    def _end_graph(self):
        self._graph.capture_end()
        self.items.append(self._graph)
        self._graph = None

    def __enter__(self):
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        self.items, self.records, self.cuts = [], [], 0
        self._pool = torch.cuda.graph_pool_handle()
        self._stream_ctx = torch.cuda.stream(torch.cuda.Stream())
        self._stream_ctx.__enter__()
        self.capturing = True
        self._begin_graph()
        return self

    def __exit__(self, exc_type, exc, tb):
        try:
            if self._graph is not None:
                self._end_graph()
        finally:
            self.capturing = False
            self._stream_ctx.__exit__(exc_type, exc, tb)
            self._stream_ctx = None
        if exc_type is not None:
            self.items = []
        return False

    def cut(self, fn):
        """Run `fn` (a collective on static buffers) NOW; while capturing, end the current graph in front of it and open
        the next one behind it, and remember `fn` for the replays."""
        if not self.capturing:
            fn()
            return
        self._end_graph()
        fn()
        torch.cuda.synchronize()     # nothing of the collective is in flight (or being polled) when the next capture opens
        self.items.append(fn)
        self.cuts += 1
        self._begin_graph()

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()

    # ---- the chained SyncBatchNorm ------------------------------------------------------------------------------------
    def sync_batch_norm(self, mod, x):
        """Training-mode forward of `mod` (a torch SyncBatchNorm subclass) on x, cut at the statistics exchange."""
        if mod.momentum is None:
            raise RuntimeError('cumulative-average SyncBatchNorm (momentum=None) reads its step counter on the host: not graph-safe')
        if len(self.records) >= MAX_CHAINED_NORMS:
            raise RuntimeError(f'more than {MAX_CHAINED_NORMS} synchronised norms in one step: use the hybrid graph mode')
        group = mod.process_group if mod.process_group is not None else dist.group.WORLD
        world = dist.get_world_size(group)
        C = x.shape[1]
        xd = _channels_first_dense(x.detach()).requires_grad_(x.requires_grad)
        with torch.no_grad():     # the statistics are constants of _ChainedNormApply; their gradient share is added in finish_backward
            if mod.track_running_stats and mod.num_batches_tracked is not None:
                mod.num_batches_tracked.add_(1)
            mean_l, invstd_l = torch.batch_norm_stats(xd, mod.eps)
            count = torch.full((1,), xd.numel() // C, dtype=mean_l.dtype, device=xd.device)
            local = torch.cat([mean_l, invstd_l, count])
            gathered = torch.empty(world, 2 * C + 1, dtype=local.dtype, device=local.device)
            if dist.get_backend(group) == 'gloo':     # no _allgather_base in gloo (torch's own SyncBatchNorm makes the same distinction)
                self.cut(lambda: dist.all_gather(list(gathered.unbind(0)), local, group=group))
            else:
                self.cut(lambda: dist.all_gather_into_tensor(gathered, local, group=group))
            mean_all, invstd_all, count_all = torch.split(gathered, C, dim=1)
            counts = count_all.reshape(-1)
            running_mean = mod.running_mean if mod.track_running_stats else None
            running_var = mod.running_var if mod.track_running_stats else None
            mean, invstd = torch.batch_norm_gather_stats_with_counts(xd, mean_all, invstd_all, running_mean, running_var,
                                                                     mod.momentum, mod.eps, counts)
        rec = _Record(mod, x, xd, mean, invstd, counts.to(torch.int32))
        self.records.append(rec)
        return _ChainedNormApply.apply(xd, mod.weight, mod.bias, mean, invstd, mod.eps, rec)

    def finish_backward(self):
        """After the backward from the loss: per chained norm, latest first -- reduce dy, exchange, input gradient, and
        continue the backward into the part of the network in front of the norm."""
        recs, self.records = self.records, []
        with torch.no_grad():
            self._finish(recs)

    def _finish(self, recs):
        for i in range(len(recs) - 1, -1, -1):
            r = recs[i]
            if r.dy is None:
                continue
            mod = r.mod
            group = mod.process_group if mod.process_group is not None else dist.group.WORLD
            dy = _channels_first_dense(r.dy)
            w = mod.weight
            need_w = w is not None and w.requires_grad
            need_b = mod.bias is not None and mod.bias.requires_grad
            sum_dy, sum_dy_xmu, gw, gb = torch.batch_norm_backward_reduce(dy, r.xd, r.mean, r.invstd, w, True, need_w, need_b)
            if need_w:
                w.grad = gw if w.grad is None else w.grad + gw
            if need_b:
                mod.bias.grad = gb if mod.bias.grad is None else mod.bias.grad + gb
            if not r.x.requires_grad:
                continue
            red = torch.cat([sum_dy, sum_dy_xmu])
            self.cut(lambda red=red, group=group: dist.all_reduce(red, group=group))
            C = sum_dy.numel()
            sum_dy, sum_dy_xmu = torch.split(red, C)
            if w is not None and w.dtype != r.mean.dtype:
                w = w.to(r.mean.dtype)
            dx = torch.batch_norm_backward_elemt(dy, r.xd, r.mean, r.invstd, w, sum_dy, sum_dy_xmu, r.counts)
            # an earlier record may share upstream nodes with this one: keep the graph until the last walk
            torch.autograd.backward(r.x, dx, retain_graph=any(q.dy is not None or q.x.requires_grad for q in recs[:i]))


def attach(model, recorder):
    """Point every ChainedSyncBatchNorm of `model` at `recorder` (None detaches).  Returns how many there are."""
    from ..layers import ChainedSyncBatchNorm
    n = 0
    for m in model.modules():
        if isinstance(m, ChainedSyncBatchNorm):
            m._segments = recorder
            n += 1
    return n

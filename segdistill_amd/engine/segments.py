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
upstream part of the autograd graph.  Two forms of the layer exist: the SegFormer head's ``linear_fuse`` tail goes through
``fused_norm_act`` (the HIP passes of csrc/batchnorm.hip: BatchNorm + ReLU + channel dropout, segdistill_amd/batchnorm.py);
any other 4-D input goes through ``sync_batch_norm``, torch.nn.SyncBatchNorm's own ATen sequence (batch_norm_stats /
gather_stats_with_counts / elemt / backward_reduce / backward_elemt).  Both combine the ranks' statistics with
torch.batch_norm_gather_stats_with_counts, so a chained step and a plain torch SyncBatchNorm step agree to rounding.
"""
from __future__ import annotations

import gc
import os
import time

import torch
import torch.distributed as dist

def quiesce_collectives():
    """Call before a hipGraph capture begins.  The process group's watchdog thread polls the completion events of outstanding
    collectives (hipEventQuery, every ~100 ms); HIP refuses that call from ANY thread while a stream captures in the default
    `global` mode (hipErrorStreamCaptureUnsupported) and the watchdog then takes the process down.  Drain the device and give
    the watchdog time to retire what has completed; captures opened by this package additionally use the `thread_local` mode,
    under which other threads are not restricted."""
    torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl':
        time.sleep(0.35)


CAPTURE_MODE = 'thread_local'
# Every chained norm costs two cuts = two more graph launches and two eager collectives per replayed step (~50 us in all).  Up to 8 that is
# always a win over the hybrid mode; a ResNet-18 student has 28 + 5 of them (~70 segments): still a whole-step replay, but the networks that
# have that many norms are bound by their 3x3 convolutions, not by launches, so the default stays conservative.  SEGDISTILL_MAX_CHAINED_NORMS
# lifts it (tests/test_segmented_graph_gpu.py runs the ResNet-18 student with 64).
MAX_CHAINED_NORMS = int(os.environ.get('SEGDISTILL_MAX_CHAINED_NORMS', '8'))


class _Record:
    """One chained norm of the step: x = its input inside the upstream autograd graph; finish(dy) -> dx (or None), the rest
    of the layer's backward including its collective, run by SegmentRecorder.finish_backward on the host thread."""
    __slots__ = ('x', 'dy', 'finish')

    def __init__(self, x, finish=None):
        self.x, self.dy, self.finish = x, None, finish


class _CollectGrad(torch.autograd.Function):
    """Runs `compute()` -- a layer's forward on a DETACHED input, cuts included -- outside autograd and returns its result; the
    backward only COLLECTS the gradient (the layer's own backward needs an all-reduce first and is run on the host thread by
    SegmentRecorder.finish_backward).  `anchor` is any tensor that requires grad, so that the output does."""

    @staticmethod
    def forward(ctx, anchor, rec, compute):
        ctx.rec = rec
        return compute()

    @staticmethod
    def backward(ctx, dy):
        rec = ctx.rec
        rec.dy = dy if rec.dy is None else rec.dy + dy
        return None, None, None


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
        self._coll_stream = None
        self.cuts = 0
        self.time_collectives = False
        self.collective_events = []

    # ---- capture / replay ---------------------------------------------------------------------------------------------
    def _begin_graph(self):
        self._graph = torch.cuda.CUDAGraph()
        self._graph.capture_begin(pool=self._pool, capture_error_mode=CAPTURE_MODE)

    def _end_graph(self):
        self._graph.capture_end()
        self.items.append(self._graph)
        self._graph = None

    def __enter__(self):
        quiesce_collectives()
        gc.collect()
        torch.cuda.empty_cache()
        self.items, self.records, self.cuts = [], [], 0
        self._pool = torch.cuda.graph_pool_handle()
        self._coll_stream = torch.cuda.Stream()
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
        from .. import deferred
        deferred.join()          # weight-gradient kernels forked to the side stream: a graph may only end with every stream joined
        if not self.capturing:
            fn()
            return
        self._end_graph()
        # At capture time the collective runs on a stream that NEVER captures, bracketed by device synchronisations (its operands are
        # not meaningful yet -- the graph in front of it was recorded, not run -- only its completion matters).  Issued on the
        # capture stream itself, its completion event would sit on a stream that is capturing again a moment later, and the process
        # group's watchdog thread, polling that event, dies with hipErrorCapturedEvent.
        torch.cuda.synchronize()
        with torch.cuda.stream(self._coll_stream):
            fn()
        quiesce_collectives()
        self.items.append(fn)
        self.cuts += 1
        self._begin_graph()

    def replay(self):
        if self.time_collectives:                      # diagnostics (bench.py's `syncbn_collectives_ms`): HIP events around every collective
            evs = []
            for it in self.items:
                if isinstance(it, torch.cuda.CUDAGraph):
                    it.replay()
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    it()
                    e1.record()
                    evs.append((e0, e1))
            self.collective_events.append(evs)
            return
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()

    def collective_ms(self):
        """Mean device time per replayed step spent in the collectives between the graphs, and how many there are (after replays with
        `time_collectives` set; synchronises)."""
        torch.cuda.synchronize()
        steps = [sum(e0.elapsed_time(e1) for e0, e1 in evs) for evs in self.collective_events if evs]
        n = len(self.collective_events[-1]) if self.collective_events else 0
        self.collective_events = []
        return (sum(steps) / len(steps) if steps else 0.0), n

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
        xd = _channels_first_dense(x.detach())
        st = {}

        def compute():
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
            st['mean'], st['invstd'] = torch.batch_norm_gather_stats_with_counts(xd, mean_all, invstd_all, running_mean, running_var,
                                                                                 mod.momentum, mod.eps, counts)
            st['counts'] = counts.to(torch.int32)
            return torch.batch_norm_elemt(xd, mod.weight, mod.bias, st['mean'], st['invstd'], mod.eps)

        def finish(dy, need_dx):
            mean, invstd = st['mean'], st['invstd']
            dy = _channels_first_dense(dy)
            w = mod.weight
            need_w = w is not None and w.requires_grad
            need_b = mod.bias is not None and mod.bias.requires_grad
            sum_dy, sum_dy_xmu, gw, gb = torch.batch_norm_backward_reduce(dy, xd, mean, invstd, w, True, need_w, need_b)
            if need_w:
                w.grad = gw if w.grad is None else w.grad + gw
            if need_b:
                mod.bias.grad = gb if mod.bias.grad is None else mod.bias.grad + gb
            if not need_dx:
                return None
            red = torch.cat([sum_dy, sum_dy_xmu])
            self.cut(lambda: dist.all_reduce(red, group=group))
            sum_dy, sum_dy_xmu = torch.split(red, C)
            if w is not None and w.dtype != mean.dtype:
                w = w.to(mean.dtype)
            return torch.batch_norm_backward_elemt(dy, xd, mean, invstd, w, sum_dy, sum_dy_xmu, st['counts'])

        rec = _Record(x, finish)
        self.records.append(rec)
        return _CollectGrad.apply(self._anchor(x, mod), rec, compute)

    @staticmethod
    def _anchor(x, mod):
        """A fresh leaf that requires grad, so that _CollectGrad's output does.  Deliberately NOT x: an edge to x's graph would
        let the backward from the loss walk into (and free) the upstream nodes that finish_backward has yet to run."""
        return torch.empty(0, device=x.device, requires_grad=True)

    def fused_norm_act(self, mod, tokens, relu, drop):
        """The HIP form (segdistill_amd/batchnorm.py): BatchNorm + ReLU + channel dropout on tokens [B, N, C], its collectives
        as cut points.  Used by the SegFormer head for its `linear_fuse` tail."""
        from .. import batchnorm as hip_bn
        if len(self.records) >= MAX_CHAINED_NORMS:
            raise RuntimeError(f'more than {MAX_CHAINED_NORMS} synchronised norms in one step: use the hybrid graph mode')
        xd = tokens.detach()
        st = {}

        def compute():
            y, st['sv'] = hip_bn.forward_pieces(xd, mod, relu, drop, cut=self.cut)
            return y

        def finish(dy, need_dx):
            dx, gw, gb = hip_bn.backward_pieces(st['sv'], dy, need_dx=need_dx, cut=self.cut)
            if mod.weight.requires_grad:
                mod.weight.grad = gw if mod.weight.grad is None else mod.weight.grad + gw
            if mod.bias.requires_grad:
                mod.bias.grad = gb if mod.bias.grad is None else mod.bias.grad + gb
            return dx

        rec = _Record(tokens, finish)
        self.records.append(rec)
        return _CollectGrad.apply(self._anchor(tokens, mod), rec, compute)

    def finish_backward(self, walk=None):
        """After the backward from the loss: per chained norm, latest first -- the layer's own backward (dy sums, exchange,
        input gradient), then the backward continues into the part of the network in front of the norm.  `walk(fn)`, when given,
        runs each of those further autograd walks (the trainer isolates them: engine/trainer.py:_isolated_walk)."""
        recs, self.records = self.records, []
        with torch.no_grad():
            for i in range(len(recs) - 1, -1, -1):
                r = recs[i]
                if r.dy is None:
                    continue
                dx = r.finish(r.dy, r.x.requires_grad)
                r.dy = r.finish = None
                if dx is not None:
                    # an earlier record may share upstream nodes with this one: keep the graph until the last walk
                    def go(x=r.x, dx=dx, keep=i > 0):
                        torch.autograd.backward(x, dx, retain_graph=keep)
                    go() if walk is None else walk(go)


def attach(model, recorder):
    """Point every ChainedSyncBatchNorm of `model` at `recorder` (None detaches).  Returns how many there are."""
    from ..layers import ChainedSyncBatchNorm
    n = 0
    for m in model.modules():
        if isinstance(m, ChainedSyncBatchNorm):
            m._segments = recorder
            n += 1
    return n

"""One KD training iteration, the way the reference's runner drives it (mmcv IterBasedRunner +
OptimizerHook: ``zero_grad -> train_step -> loss.backward -> step``, reference
mmseg/apis/train.py:97-138 and SURVEY.md Appendix B), on the MI355X-first DP substrate."""
from __future__ import annotations

import contextlib

import torch

from .. import deferred
from . import segments
from .dp import DataParallelReducer
from .optim import PolyLR, build_optimizer, sync_derived_weights


def _issues_memsets(fn):
    """True if running `fn` enqueues any hipMemset command (seen through the torch profiler).

    Why it matters (ROCm 7.x runtime bundled with PyTorch 2.10): a memset NODE inside a
    replayed hipGraph takes its fill pattern from a staging area shared with every other memset on the device; a
    hipMemsetAsync issued on ANOTHER stream while the node is pending makes the node write garbage (kernel-argument
    words of the other memset) instead of its value.  Two pieces of work may therefore only overlap on different
    streams if at most one of them contains memsets.  The MiT/ResNet/Swin teacher forward contains none (checked here
    at capture time); the student's fwd+bwd contained ~30 in round 1 (zero-initialised autograd buffers) and none since round 6 (segmentors/base.py::_mean
    removed the last one, aten::mean's semaphore clear)."""
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return any('emset' in r.key for r in prof.key_averages())


class _TupleOut(torch.nn.Module):
    """make_graphed_callables wants tuple outputs; MiT / ResNet backbones return a list or tuple of feature maps."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, x):
        return tuple(self.inner(x))


def _freeze_dead_parameters(model):
    """SegFormerHead never uses the conv_seg it inherits: keep it in the state dict,
    keep it out of the optimizer / reducer."""
    from ..decode_heads import SegFormerHead
    for m in model.modules():
        if isinstance(m, SegFormerHead):
            for p in m.conv_seg.parameters():
                p.requires_grad = False


class KDTrainer:
    def __init__(self, model, optimizer_cfg, lr_cfg=None, max_iters=160000, world=1, log_interval=50, precision=None):
        self.model = model
        _freeze_dead_parameters(model)
        self.optimizer = build_optimizer(model, optimizer_cfg)
        self.reducer = DataParallelReducer([p for g in self.optimizer.param_groups for p in g['params']], world=world)
        self.reducer.broadcast_parameters(model)
        if lr_cfg is None:
            self.sched = None                       # constant learning rate
        else:
            lr_cfg = dict(lr_cfg)
            policy = lr_cfg.pop('policy', 'poly')
            if policy != 'poly':
                raise NotImplementedError(f"lr_config policy {policy!r}: every KD config of the reference uses 'poly' (schedule_160k_adamw.py)")
            lr_cfg.pop('by_epoch', None)
            self.sched = PolyLR(self.optimizer, max_iters=max_iters, **lr_cfg)
        self.iter = 0
        self.log_interval = log_interval
        self.last_log_vars = None
        model.defer_log_sync = True  # host sync only when a log line is due
        # precision=dict(activations='bf16'): bf16 storage of activations / tapped features (autocast), fp32 master weights,
        # fp32 accumulation inside every HIP kernel (BASELINE config 5)
        self.bf16 = bool(precision) and precision.get('activations') == 'bf16'
        if self.bf16 and hasattr(model, 'activation_dtype'):
            model.activation_dtype = torch.bfloat16     # the teacher forward opens its own autocast region (sd_module.py)

    # ---- hipGraph mode ------------------------------------------------------------------------------------------------
    # The KD step is ~1300 kernel launches; eager host enqueue costs ~24 ms/step on the GPU box (round-1 host probe),
    # about as much as the GPU work itself.  Forward + backward are therefore captured ONCE into a hipGraph (static
    # input buffers; the teacher's side stream forks/joins inside the capture) and replayed; everything that changes
    # per iteration reaches the kernels as data (alpha scalar, permutation table -- distillation/losses.py), the
    # gradient all-reduce and the fused optimizer step stay outside the graph.  With more than one rank the student's
    # SyncBatchNorm exchanges statistics mid-forward and mid-backward: the capture is CUT there (engine/segments.py) and
    # the step becomes a chain graph | all_gather | graph | all_reduce | graph -- no RCCL call is ever recorded.
    def enable_graph(self, example_batch):
        """Capture the step for batches shaped like `example_batch` as TWO graphs: the frozen teacher's forward (side
        stream; replayed one iteration AHEAD when the next batch is known, so it overlaps the current backward) and the
        student's forward + losses + backward (main stream; reads the teacher taps from static buffers).  Returns True
        on success; on any failure the trainer stays in eager mode (and says why)."""
        import warnings
        if not example_batch['img'].is_cuda:
            return False
        m = self.model
        if getattr(m, 'log_grad', False):   # the gradient-angle diagnostic back-propagates twice inside train_step: eager only
            warnings.warn('hipGraph capture skipped: distillation[0] asks for log_grad')
            return False
        try:
            self._static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in example_batch.items()}
            has_kd = hasattr(m, 'distillation_loss') and bool(getattr(m, 'distillation', None))
            if hasattr(m, 'distillation_loss'):
                m.distillation_loss.set_graph_safe(True)
            m.train()
            self._seg = segments.SegmentRecorder()
            segments.attach(m, self._seg)   # synchronised norms (world > 1) become cut points of the capture
            cnt0 = getattr(m, 'cnt', 0)
            buffers0 = self._snapshot_buffers()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):   # warm-up on a non-default stream, as graph capture requires
                for _ in range(2):
                    self.reducer.zero_grad()
                    self._fwd_bwd(self._static)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if hasattr(m, 'cnt'):
                m.cnt = cnt0                # the warm-up passes do not count as training iterations ...
            self._restore_buffers(buffers0)  # ... and must not advance the student's BatchNorm running statistics either
            if has_kd:
                # (1) teacher graph on the model's side stream; its tapped features are static outputs
                ts = m._side_stream or torch.cuda.Stream(device=example_batch['img'].device)
                m._side_stream = ts
                self._t_img = example_batch['img'].clone()
                m._prefetched = None
                m.extractor.teacher_features.clear()
                gt = torch.cuda.CUDAGraph()
                segments.quiesce_collectives()      # the warm-up passes may have left collectives for the watchdog to retire
                with torch.cuda.graph(gt, stream=ts, capture_error_mode=segments.CAPTURE_MODE):
                    m._teacher_forward(self._t_img, None, None)
                self._t_out = dict(m.extractor.teacher_features)
                m.extractor.clear()
                if not self._t_out:
                    raise RuntimeError('the teacher capture produced no tapped feature')
                self._t_graph = gt
                # overlap teacher(k+1) with the student step k only if the teacher forward issues no memset (see _issues_memsets)
                self._overlap_teacher = not _issues_memsets(lambda: m._teacher_forward(self._t_img, None, None))
                m.extractor.teacher_features.clear()
                if not self._overlap_teacher:
                    warnings.warn('the teacher forward issues hipMemset commands: its graph will not be overlapped with the student graph')
                self._t_cur = {k: torch.empty_like(v) for k, v in self._t_out.items()}
                m._taps_override = self._t_cur   # the captured student step reads the taps from these buffers
                self._primed = None
            m.external_step = True
            self.reducer.zero_grad()        # .grad = None: the captured backward creates the (static) gradient tensors
            with self._seg:
                self._graph_out = self._fwd_bwd(self._static)
                if self.reducer.collective:
                    # the pack into the flat all-reduce buffer is recorded too: it reads the capture's STATIC gradient tensors
                    # (after the first exchange `.grad` points at the flat views, which a replay does not write)
                    self.reducer.pack()
            self._graph_packs = self.reducer.collective
            self._graph = self._seg         # .replay(): the graphs in capture order, the collectives between them
            segments.attach(m, None)
            torch.cuda.synchronize()
            return True
        except Exception as e:  # noqa: BLE001 -- any capture problem must degrade to eager, not kill the run
            warnings.warn(f'hipGraph capture failed ({type(e).__name__}: {e}); continuing in eager mode')
            self.graph_error = f'{type(e).__name__}: {e}'     # for callers that must say why (bench.py's line, the tests)
            if hasattr(m, 'cnt') and 'cnt0' in locals():
                m.cnt = cnt0
            self.disable_graph()
            return False

    def disable_graph(self):
        """Back to eager stepping: after a failed capture, or when ANOTHER rank's capture failed -- every rank of a data-parallel job must step
        the same way (bench.agree_on_graph_mode): a rank replaying graphs next to eager ranks would still issue the same collectives, but a
        half-captured rank must not be left with static buffers the others do not have."""
        m = self.model
        self._graph = None
        self._seg = None
        self._t_graph = None
        if getattr(m, '_graphed_teacher', None) is not None:      # hybrid mode's pieces
            m._graphed_teacher = None
        if hasattr(m, 'student') and getattr(m.student, '_graphed_backbone', None) is not None:
            object.__setattr__(m.student, '_graphed_backbone', None)
        segments.attach(m, None)
        m.external_step = False
        if hasattr(m, '_taps_override'):
            m._taps_override = None
        if hasattr(m, '_prefetched'):
            m._prefetched = None
        if hasattr(m, 'distillation_loss'):
            m.distillation_loss.set_graph_safe(False)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    # ---- hybrid graph mode (safe at any world size) -----------------------------------------------------------------------
    # The full-step capture above would have to record the student's SyncBatchNorm collectives (RCCL inside a hipGraph
    # capture) when world > 1.  The hybrid mode captures only collective-free pieces, which still hold most of the launches:
    #   * the frozen teacher's forward (one forward-only graph on the side stream; its tapped features are static outputs);
    #   * the student BACKBONE's forward and backward (torch.cuda.make_graphed_callables: replayed from inside eager autograd).
    # The SegFormer/PSP head (SyncBN), the loss kernels, the gradient all-reduce and the optimizer stay eager.
    def enable_hybrid_graph(self, example_batch):
        import warnings
        m = self.model
        img = example_batch['img']
        if not img.is_cuda or not hasattr(m, 'student'):
            return False
        try:
            tapped_s = list(m.extractor.hooked['student']) if hasattr(m, 'extractor') else []
            if any(n.startswith('backbone') for n in tapped_s):
                raise RuntimeError('a student tap lives inside the backbone: its hook would not fire during graph replay')
            m.train()
            cnt0 = m.cnt
            buffers0 = self._snapshot_buffers()     # make_graphed_callables' warm-up iterations run the backbone in training mode
            h_img = img.clone()
            # graphed backward nodes run on the capture side stream while AccumulateGrad lives on the main one: benign here
            try:
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
            except AttributeError:
                pass
            # (1) student backbone: forward + backward graphs, replayed from inside eager autograd -- unless it holds synchronised
            # norms (ResNet students with ranks > 1): their collectives must not be recorded (over gloo the attempt invalidates
            # the capture and leaves the HIP runtime unusable), so such a backbone stays eager and only the teacher is graphed
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            sync_norms = multi and any(isinstance(mod, torch.nn.SyncBatchNorm) for mod in m.student.backbone.modules())
            segments.quiesce_collectives()          # earlier eager steps may have left collectives for the watchdog to retire
            if not sync_norms:
                wrapper = _TupleOut(m.student.backbone)
                wrapper.train()
                with self._autocast(cache_enabled=False):   # graphed callables must not share autocast's weight-cast cache
                    torch.cuda.make_graphed_callables(wrapper, (h_img,), num_warmup_iters=2)
                object.__setattr__(m.student, '_graphed_backbone', wrapper)   # not registered as a sub-module (state dict unchanged)
            # (2) teacher: forward-only graph on the side stream; its tapped features are static outputs
            side = m._side_stream or torch.cuda.Stream(device=img.device)
            m._side_stream = side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    m._teacher_forward(h_img, None, None)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            m.extractor.teacher_features.clear()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode=segments.CAPTURE_MODE):
                m._teacher_forward(h_img, None, None)
            outs = dict(m.extractor.teacher_features)
            m.extractor.clear()
            if not outs:
                raise RuntimeError('the teacher capture produced no tapped feature')
            m._graphed_teacher = (g, outs, h_img)
            m.prefetch_ok = not _issues_memsets(lambda: m._teacher_forward(h_img, None, None))
            m.extractor.teacher_features.clear()
            m.cnt = cnt0
            torch.cuda.synchronize()
            self._restore_buffers(buffers0)
            return True
        except Exception as e:  # noqa: BLE001
            warnings.warn(f'hybrid hipGraph capture failed ({type(e).__name__}: {e}); continuing in eager mode')
            self.graph_error = f'{type(e).__name__}: {e}'
            m._graphed_teacher = None
            if hasattr(m, 'student') and getattr(m.student, '_graphed_backbone', None) is not None:
                object.__setattr__(m.student, '_graphed_backbone', None)
            torch.cuda.synchronize()
            return False

    def _trainable_net(self):
        return self.model.student if hasattr(self.model, 'student') else self.model

    def _snapshot_buffers(self):
        return [b.detach().clone() for b in self._trainable_net().buffers()]

    def _restore_buffers(self, saved):
        with torch.no_grad():
            for b, b0 in zip(self._trainable_net().buffers(), saved):
                b.copy_(b0)

    def _autocast(self, cache_enabled=True):
        if self.bf16 and torch.cuda.is_available():
            return torch.autocast('cuda', dtype=torch.bfloat16, cache_enabled=cache_enabled)
        return contextlib.nullcontext()

    def _fwd_bwd(self, batch):
        with (self._autocast() if batch['img'].is_cuda else contextlib.nullcontext()):
            out = self.model.train_step(batch, self.optimizer)
        self._backward(out['loss'])
        return out

    def _backward(self, loss):
        """loss.backward() with the parameter-gradient combines of the HIP ops deferred to ONE launch at its end (deferred.py).

        A deferred gradient is a view of a buffer that is only WRITTEN when its scope ends, so inside one scope no parameter may
        receive two gradients (AccumulateGrad would add the second into the unwritten first).  With chained SyncBatchNorm layers the
        backward is several autograd walks -- from the loss to the norms, then from each norm's input further up (segments.py) -- and a
        KD tap upstream of a norm (config 5 taps decode_head.linear_c1..4, in front of linear_fuse's norm) puts the same parameters
        into two walks.  Every walk therefore gets its own scope and starts from `.grad = None`; gradients of earlier walks are set
        aside and added back (one multi-tensor add, and only for parameters that really were reached twice)."""
        cuda = loss.is_cuda
        seg = getattr(self, '_seg', None)
        # later walks may pass through nodes this one has already visited (a KD tap upstream of a chained norm shares the backbone with the
        # path through the norm): keep the graph until the last walk (finish_backward frees it there)
        keep = seg is not None and bool(seg.records)
        with (deferred.scope() if cuda else contextlib.nullcontext()):
            loss.backward(retain_graph=keep)
        if seg is not None and seg.records:
            seg.finish_backward(walk=self._isolated_walk if cuda else None)

    def _isolated_walk(self, fn):
        params = self.reducer.params
        held = [(p, p.grad) for p in params if p.grad is not None]
        for p, _ in held:
            p.grad = None
        with deferred.scope():
            fn()
        twice = [(p, g) for p, g in held if p.grad is not None]
        if twice:
            torch._foreach_add_([p.grad for p, _ in twice], [g for _, g in twice])
        for p, g in held:
            if p.grad is None:
                p.grad = g

    def _replay_teacher(self, img):
        """side stream: copy the image into the teacher graph's input and replay it; returns the completion event."""
        side = self.model._side_stream
        with torch.cuda.stream(side):
            self._t_img.copy_(img, non_blocking=True)
            self._t_graph.replay()
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    def _graph_step(self, batch, next_batch=None):
        m = self.model
        main = torch.cuda.current_stream()
        if getattr(self, '_t_graph', None) is not None:
            img = batch['img']
            if self._primed is not None and self._primed[0] is img and self._primed[1] == img._version:
                ev = self._primed[2]                                   # replayed during the previous iteration (same tensor object, unmodified)
            else:
                m._side_stream.wait_stream(main)
                ev = self._replay_teacher(img)
            main.wait_event(ev)
            for k, v in self._t_out.items():                           # free the teacher graph's outputs for the next replay
                self._t_cur[k].copy_(v, non_blocking=True)
            self._primed = None
            if next_batch is not None and self._overlap_teacher:
                copied = torch.cuda.Event()
                copied.record(main)
                m._side_stream.wait_event(copied)
                nimg = next_batch['img']
                self._primed = (nimg, nimg._version, self._replay_teacher(nimg))   # overlaps everything below
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                self._static[k].copy_(v, non_blocking=True)
        m.cnt += 1
        if hasattr(m, 'distillation_loss'):
            m.distillation_loss.prepare_replay(m.cnt)
        self._graph.replay()
        return self._graph_out

    def step(self, batch, next_batch=None):
        """One training iteration.  `next_batch` (optional) lets the frozen teacher's forward for the NEXT iteration be
        launched now, overlapping this iteration's backward (the teacher never depends on the optimizer step)."""
        self.model.train()
        if self.sched is not None:
            self.sched.step(self.iter)
        if getattr(self, '_graph', None) is not None:
            out = self._graph_step(batch, next_batch)
        else:
            self.reducer.zero_grad()
            with (self._autocast() if batch['img'].is_cuda else contextlib.nullcontext()):
                out = self.model.train_step(batch, self.optimizer)
            if next_batch is not None and hasattr(self.model, 'prefetch_teacher'):
                self.model.prefetch_teacher(next_batch['img'])
            self._backward(out['loss'])
        if getattr(self, '_graph', None) is not None and self._graph_packs:
            self.reducer.exchange()         # packed by the replayed graph
        else:
            self.reducer.all_reduce()
        self.optimizer.step()             # a torch optimizer rewrites planes / shadows through its step post-hook (engine/optim.py)
        self.iter += 1
        self.last_log_vars = out['log_vars']
        return out

    # ---- checkpoint / resume -----------------------------------------------------------------------------------------
    # The reference checkpoints the whole SDModule including the frozen teacher (mmcv CheckpointHook) and loses the
    # distillation step counter on resume (warm-up / early-decay / shuffle schedules restart).  Here the
    # file keeps mmcv's layout -- {'meta', 'state_dict', 'optimizer'} -- with 'state_dict' = the STUDENT segmentor's own keys
    # (backbone.*, decode_head.*: what the reference's test / inference tooling and this repo's load_checkpoint / `s_pretrain`
    # load straight into an EncoderDecoder), meta = {'iter', 'cnt'}, plus the trainable align projections.
    def state_dict(self):
        m = self.model
        sd = {'meta': {'iter': self.iter, 'cnt': getattr(m, 'cnt', self.iter)}, 'optimizer': self.optimizer.state_dict()}
        if hasattr(m, 'student'):
            sd['state_dict'] = m.student.state_dict()
            sd['distillation_loss'] = m.distillation_loss.state_dict()
        else:
            sd['state_dict'] = m.state_dict()
        return sd

    def load_state_dict(self, sd):
        m = self.model
        weights = sd['state_dict'] if 'state_dict' in sd else sd.get('student', sd.get('model'))   # 'student' / 'model': round-1 files
        if hasattr(m, 'student'):
            m.student.load_state_dict(weights)
            m.distillation_loss.load_state_dict(sd.get('distillation_loss', {}), strict=False)
        else:
            m.load_state_dict(weights)
        self.optimizer.load_state_dict(sd['optimizer'])
        self._sync_derived_weights()
        meta = sd.get('meta', sd)
        self.iter = int(meta['iter'])
        if hasattr(m, 'cnt'):
            m.cnt = int(meta['cnt'])

    def _sync_derived_weights(self):
        """Copies of the weights that kernels read instead of the fp32 parameter are rewritten IN PLACE now (engine/optim.py::
        sync_derived_weights): a captured graph has their addresses baked in and a replay runs no Python forward."""
        sync_derived_weights(self.reducer.params)

    def save(self, path):
        import os
        tmp = path + '.tmp'
        torch.save(self.state_dict(), tmp)
        os.replace(tmp, path)

    def resume(self, path, map_location=None):
        self.load_state_dict(torch.load(path, map_location=map_location, weights_only=False))

    def log_values(self):
        """Host copies of the most recent log variables, averaged over the ranks: one packed all-reduce + one device->host
        sync.  COLLECTIVE: every rank must call it at the same iterations (the model defers the per-step log reduction of
        the reference, base.py:204-207, to this point)."""
        if self.last_log_vars is None:
            return {}
        names = list(self.last_log_vars)
        packed = torch.stack([torch.as_tensor(self.last_log_vars[n]).float().reshape(()) for n in names])
        if self.reducer.world > 1 and getattr(self.model, 'defer_log_sync', False):
            import torch.distributed as dist
            packed = packed.to(self.reducer.flat.device) / self.reducer.world
            dist.all_reduce(packed)
        return dict(zip(names, packed.tolist()))

"""Every hand-written kernel of the KD step at its BASELINE shape, through the C ABI, timed with HIP events on the stream the
kernels are launched on, against the roofline that bounds it.

    python tools/kernel_rooflines.py [--only align,r2,...] [--reps 20] [--json out.json]

Standalone it is also the program the rocprofv3 passes of tools/refresh_profiles.sh run (kernel trace; FETCH_SIZE / WRITE_SIZE /
SQ_* counters in separate --pmc passes).  bench.py imports `run()` for the `roofline.kernels` list of its JSON line.

Peaks (MI355X_MICROARCH.md): HBM 8.0 TB/s; f32-input MFMA 157.3 TF (dense, = the vector rate); bf16 MFMA 2500 TF dense;
VALU 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-instructions/s (a transcendental counts 4: quarter rate).
Algorithmic work per launch is stated per entry (`work`), so achieved = work / time is checkable."""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM = 8000.0          # GB/s
MFMA_F32 = 157.3      # TFLOP/s, v_mfma_f32_32x32x2_f32
MFMA_BF16 = 2500.0    # TFLOP/s dense
VALU = 78.6           # T lane-instructions / s


def _time(fn, reps, per_graph=5, graph=True):
    """DEVICE ms per call: `per_graph` calls captured into a hipGraph, the graph replayed `reps` times, HIP events around each replay on the
    stream the kernels run on (median).  An eager loop would measure the host for the kernels that run under ~20 us."""
    fn0 = fn

    def fn():
        return fn0(torch.cuda.current_stream().cuda_stream)      # the stream current at CALL time: the capture below runs on a side stream

    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    try:
        if not graph:
            raise RuntimeError('eager timing requested')
        with torch.cuda.stream(side):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                for _ in range(per_graph):
                    fn()
        run = g.replay
    except Exception:  # noqa: BLE001 -- an op that cannot be captured: time the eager loop instead
        torch.cuda.synchronize()

        def run():
            for _ in range(per_graph):
                fn()
    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        run()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2] / per_graph


def _entry(name, kernels, shape, dtype, ms, bound, work, unit_peak, note=None):
    """work: algorithmic bytes (hbm) / flops (mfma) / lane-instructions (valu) of ONE launch."""
    if bound == 'hbm':
        ach, unit = work / (ms * 1e-3) / 1e9, 'GB/s'
    elif bound == 'mfma':
        ach, unit = work / (ms * 1e-3) / 1e12, 'TFLOP/s'
    else:
        ach, unit = work / (ms * 1e-3) / 1e12, 'T lane-inst/s'
    e = {'name': name, 'kernels': kernels, 'shape': shape, 'dtype': dtype, 'ms': round(ms, 4), 'bound': bound, 'work': int(work),
         'achieved': round(ach, 2), 'peak': unit_peak, 'unit': unit, 'frac': round(ach / unit_peak, 4)}
    if note:
        e['note'] = note
    return e


def _ok(rc, what):
    from segdistill_amd import _lib
    _lib.check(rc, what)


# ---------------------------------------------------------------------------------------------------------------------------
def bench_r1(dev, reps, B=8, C=150, HW=512, g=8, tau=4.0, dtype=torch.float32):
    from segdistill_amd import _lib
    L = _lib.lib()
    DT = 0 if dtype == torch.float32 else 1
    gen = torch.Generator(device=dev).manual_seed(1234)
    S = (2 * torch.randn(B, C, HW, HW, device=dev, generator=gen)).to(dtype)
    T = (2 * torch.randn(B, C, HW, HW, device=dev, generator=gen)).to(dtype)
    rows = B * (-(-C // g))
    lse, kl, loss, dS = torch.empty(rows, 2, device=dev), torch.empty(rows, device=dev), torch.empty((), device=dev), torch.empty_like(S)
    wsb = L.sd_cgd_kl_workspace_bytes(B, C, HW, HW, g)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    tf = _time(lambda st: _ok(L.sd_cgd_kl_fwd(S.data_ptr(), T.data_ptr(), DT, B, C, HW, HW, g, 1 / tau, 3.0 / rows, None, lse.data_ptr(), kl.data_ptr(),
                                            loss.data_ptr(), ws.data_ptr(), wsb, st), 'fwd'), reps)
    tb = _time(lambda st: _ok(L.sd_cgd_kl_bwd(S.data_ptr(), T.data_ptr(), DT, B, C, HW, HW, g, 1 / tau, 3.0 / (rows * tau), None, lse.data_ptr(), None,
                                            dS.data_ptr(), st), 'bwd'), reps)
    N, e = S.numel(), S.element_size()
    tag = 'f32' if dtype == torch.float32 else 'bf16'
    return [_entry(f'cgd_kl R1 fwd ({tag})', 'cgd_fwd_partials + cgd_fwd_rows + cgd_fwd_loss', [B, C, HW, HW], tag, tf, 'hbm', 2 * N * e, HBM),
            _entry(f'cgd_kl R1 bwd ({tag})', 'cgd_bwd', [B, C, HW, HW], tag, tb, 'hbm', 3 * N * e, HBM)]


def bench_tok(dev, reps, B=8, C=768, g=8, tau=4.0, dtype=torch.bfloat16):
    """Token-major criterion at the four config-5 stage shapes [8, h*h, 768] (bf16 taps), h = 128, 64, 32, 16: each stage as its own call
    (2 launches forward: scan + finish; 1 backward) and -- what the train step issues since round 4 -- stages 2-4 and all four stages as ONE
    call (sd_cgd_kl_tok_fwd_multi / _bwd_multi)."""
    import ctypes as C_
    from segdistill_amd import _lib
    from segdistill_amd.ops import _TokBwdJob, _TokFwdJob
    L = _lib.lib()
    DT = 0 if dtype == torch.float32 else 1
    tag = 'bf16' if dtype == torch.bfloat16 else 'f32'
    out, st_ = [], []
    for h in (128, 64, 32, 16):
        P = h * h
        gen = torch.Generator(device=dev).manual_seed(1234)
        S = (2 * torch.randn(B, P, C, device=dev, generator=gen)).to(dtype)
        T = (2 * torch.randn(B, P, C, device=dev, generator=gen)).to(dtype)
        rows = B * (-(-C // g))
        lse, kl, loss, dS = torch.empty(rows, 2, device=dev), torch.empty(rows, device=dev), torch.empty((), device=dev), torch.empty_like(S)
        wsb = L.sd_cgd_kl_tok_workspace_bytes(B, C, P)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        st_.append((S, T, lse, kl, loss, dS, ws, wsb, P, rows))
        tf = _time(lambda st: _ok(L.sd_cgd_kl_tok_fwd(S.data_ptr(), T.data_ptr(), DT, B, C, P, g, 1 / tau, 3.0 / rows, None, lse.data_ptr(), kl.data_ptr(),
                                                      loss.data_ptr(), ws.data_ptr(), wsb, st), 'tok fwd'), reps)
        tb = _time(lambda st: _ok(L.sd_cgd_kl_tok_bwd(S.data_ptr(), T.data_ptr(), DT, B, C, P, g, 1 / tau, 3.0 / (rows * tau), None, lse.data_ptr(), None,
                                                      dS.data_ptr(), st), 'tok bwd'), reps)
        N, e = S.numel(), S.element_size()
        note = None if h >= 64 else 'operands fit the 256 MiB Infinity Cache / launch-bound at this size'
        out += [_entry(f'cgd_kl token-major fwd, cfg5 stage {(128, 64, 32, 16).index(h) + 1} ({tag})', 'cgd_tok_fwd_partials + cgd_tok_finish',
                       [B, P, C], tag, tf, 'hbm', 2 * N * e, HBM, note),
                _entry(f'cgd_kl token-major bwd, cfg5 stage {(128, 64, 32, 16).index(h) + 1} ({tag})', 'cgd_tok_bwd', [B, P, C], tag, tb, 'hbm', 3 * N * e, HBM, note)]
    for name, sel in (('stages 2-4 in one call', st_[1:]), ('all four stages in one call', st_)):
        fj, bj = (_TokFwdJob * len(sel))(), (_TokBwdJob * len(sel))()
        nbytes = 0
        for i, (S, T, lse, kl, loss, dS, ws, wsb, P, rows) in enumerate(sel):
            f, b = fj[i], bj[i]
            f.S, f.T, f.perm, f.row_lse2, f.row_kl, f.loss, f.workspace, f.workspace_bytes = S.data_ptr(), T.data_ptr(), None, lse.data_ptr(), kl.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb
            f.P, f.B, f.C, f.g, f.inv_tau, f.loss_scale = P, B, C, g, 1 / tau, 3.0 / rows
            b.S, b.T, b.perm, b.row_lse2, b.upstream, b.dS = S.data_ptr(), T.data_ptr(), None, lse.data_ptr(), None, dS.data_ptr()
            b.P, b.B, b.C, b.g, b.inv_tau, b.coef = P, B, C, g, 1 / tau, 3.0 / (rows * tau)
            nbytes += S.numel() * S.element_size()
        tf = _time(lambda st: _ok(L.sd_cgd_kl_tok_fwd_multi(C_.cast(fj, C_.c_void_p), len(sel), DT, st), 'tok fwd multi'), reps)
        tb = _time(lambda st: _ok(L.sd_cgd_kl_tok_bwd_multi(C_.cast(bj, C_.c_void_p), len(sel), DT, st), 'tok bwd multi'), reps)
        shape = [[B, q[8], C] for q in sel]
        out += [_entry(f'cgd_kl token-major fwd, cfg5 {name} ({tag})', 'cgd_tok_fwd_partials (x1-2) + cgd_tok_finish', shape, tag, tf, 'hbm', 2 * nbytes, HBM),
                _entry(f'cgd_kl token-major bwd, cfg5 {name} ({tag})', 'cgd_tok_bwd', shape, tag, tb, 'hbm', 3 * nbytes, HBM)]
    return out


def bench_r2(dev, reps, B=8, C=150, hw=128, F=4, g=8, tau=4.0):
    """VALU-bound: per OUTPUT element and tensor ~ (1 + 3/F) interpolation FMAs + the online-softmax fold (~5 VALU + 1.06 exp for the pair)."""
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(1234)
    s = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    t = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    H = hw * F
    rows = B * (-(-C // g))
    lse, kl, loss, ds = torch.empty(rows, 2, device=dev), torch.empty(rows, device=dev), torch.empty((), device=dev), torch.empty_like(s)
    wsb = L.sd_cgd_kl_up_workspace_bytes(B, C, hw, hw, H, H, g)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    tf = _time(lambda st: _ok(L.sd_cgd_kl_up_fwd(s.data_ptr(), t.data_ptr(), 0, B, C, hw, hw, H, H, g, 1 / tau, 3.0 / rows, None, lse.data_ptr(),
                                               kl.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, st), 'fwd'), reps)
    tb = _time(lambda st: _ok(L.sd_cgd_kl_up_bwd(s.data_ptr(), t.data_ptr(), 0, B, C, hw, hw, H, H, g, 1 / tau, 3.0 / (rows * tau), None,
                                               lse.data_ptr(), None, ds.data_ptr(), st), 'bwd'), reps)
    N = B * C * H * H
    # lane-instruction model per output element (both tensors), see DESIGN.md section 3.2: forward 2*(1.75 lerp FMA) + fold 9 + 2.125 exp*4
    fwd_ops = N * (2 * 1.75 + 9 + 2.125 * 4)
    bwd_ops = N * (2 * 1.75 + 6 + 2 * 4 + 2.0)      # recompute lerps, two exponentials, dS, transposed-interpolation FMAs
    note = 'exp/VALU-bound by design (reads only the taps: 5*B*C*h*w*e bytes); lane-instruction count is a MODEL, the measured SQ_INSTS_VALU is in profiles/'
    out = [_entry('cgd_kl R2 fwd (fused x4 upsample)', 'cgd_up_fwd_partials (+ rows, loss)', [B, C, hw, hw, '->', H, H], 'f32', tf, 'valu', fwd_ops, VALU, note),
           _entry('cgd_kl R2 bwd (fused x4 upsample)', 'cgd_up_bwd', [B, C, hw, hw, '->', H, H], 'f32', tb, 'valu', bwd_ops, VALU, note)]
    # config 3: the CGD criterion (g = 8, tau = 4) and the channel-wise KL (g = 1, tau = 1) on the same taps in ONE pass each way
    rows_b = B * C
    lse_b, kl_b, loss_b = torch.empty(rows_b, 2, device=dev), torch.empty(rows_b, device=dev), torch.empty((), device=dev)
    ws2 = torch.empty(2 * wsb, dtype=torch.uint8, device=dev)
    tf2 = _time(lambda st: _ok(L.sd_cgd_kl_up_fwd2(s.data_ptr(), t.data_ptr(), 0, B, C, hw, hw, H, H, None, g, 1 / tau, 3.0 / rows, lse.data_ptr(), kl.data_ptr(),
                                                 loss.data_ptr(), 1, 1.0, 1.0 / rows_b, lse_b.data_ptr(), kl_b.data_ptr(), loss_b.data_ptr(), ws2.data_ptr(), 2 * wsb, st),
                               'fwd2'), reps)
    tb2 = _time(lambda st: _ok(L.sd_cgd_kl_up_bwd2(s.data_ptr(), t.data_ptr(), 0, B, C, hw, hw, H, H, None, g, 1 / tau, 3.0 / (rows * tau), lse.data_ptr(), None,
                                                 1, 1.0, 1.0 / rows_b, lse_b.data_ptr(), None, ds.data_ptr(), st), 'bwd2'), reps)
    note2 = note + '; TWO criteria per pass: the fold / the exponentials are per criterion, loads and interpolation are shared'
    out += [_entry('cgd_kl R2 fwd, two criteria in one pass (config 3: g=8 tau=4 + g=1 tau=1)', 'cgd_up_fwd_partials<DUAL> (+ 2 x rows, loss)',
                   [B, C, hw, hw, '->', H, H], 'f32', tf2, 'valu', N * (2 * 1.75 + 2 * (9 + 2.125 * 4)), VALU, note2),
            _entry('cgd_kl R2 bwd, two criteria in one pass (config 3)', 'cgd_up_bwd<DUAL>', [B, C, hw, hw, '->', H, H], 'f32', tb2, 'valu',
                   N * (2 * 1.75 + 6 + 4 * 4 + 2.0 + 2.0), VALU, note2)]
    return out


def bench_align(dev, reps, B, Cs, Ct, h, dtype, tag):
    from segdistill_amd import _lib
    L = _lib.lib()
    DT = 0 if dtype == torch.float32 else 1
    gen = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(B, Cs, h, h, device=dev, generator=gen).to(dtype)
    w = torch.randn(Ct, Cs, device=dev, generator=gen) * 0.05
    b = torch.randn(Ct, device=dev, generator=gen) * 0.05
    y = torch.empty(B, Ct, h, h, device=dev, dtype=dtype)
    dy = torch.randn(B, Ct, h, h, device=dev, generator=gen).to(dtype)
    dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    wsb = L.sd_align1x1_workspace_bytes(B, Cs, Ct, h, h)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    flops = 2.0 * Ct * Cs * B * h * h
    e = x.element_size()
    shape = {'B': B, 'Cs': Cs, 'Ct': Ct, 'h': h, 'w': h, 'gemm': f'M={Ct} K={Cs} N={B * h * h}'}
    out = []
    # fp32 storage has two arithmetic modes (tunable align_split_bf16): split-bf16 on the bf16 matrix pipe (shipped) and exact f32 MFMA
    modes = ((1, 'split-bf16', MFMA_BF16 / 6), (0, 'f32 MFMA', MFMA_F32)) if dtype == torch.float32 else ((None, 'bf16', MFMA_BF16),)
    for mode, mtag, peak in modes:
        if mode is not None:
            _lib.set_tunable('align_split_bf16', mode)
        tf = _time(lambda st: _ok(L.sd_align1x1_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), DT, B, Cs, Ct, h, h, st), 'fwd'), reps)
        td = _time(lambda st: _ok(L.sd_align1x1_bwd_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), DT, B, Cs, Ct, h, h, st), 'bwd_data'), reps)
        tw = _time(lambda st: _ok(L.sd_align1x1_bwd_weight(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), DT, B, Cs, Ct, h, h, ws.data_ptr(), wsb, st),
                               'bwd_weight'), reps)
        for nm, ms, byt in (('fwd', tf, (x.numel() + y.numel()) * e), ('bwd_data', td, (dy.numel() + dx.numel()) * e), ('bwd_weight', tw, (dy.numel() + x.numel()) * e)):
            ent = _entry(f'align1x1 {nm} {tag} ({mtag})' if mode is not None else f'align1x1 {nm} {tag}', f'sd_align1x1_{nm}', shape,
                         'f32' if dtype == torch.float32 else 'bf16', ms, 'mfma', flops, round(peak, 1),
                         'peak = dense bf16 MFMA / 6 cross products; %.0f %% of the f32-input MFMA peak' % (100 * flops / (ms * 1e-3) / 1e12 / MFMA_F32) if mode == 1 else None)
            ent['hbm_GBps'] = round(byt / (ms * 1e-3) / 1e9, 1)
            ent['hbm_frac'] = round(ent['hbm_GBps'] / HBM, 4)
            ent['algorithmic_bytes'] = int(byt)
            out.append(ent)
    if dtype == torch.float32:
        _lib.set_tunable('align_split_bf16', 1)
    return out


def bench_sra(dev, reps, B=8):
    """MiT spatial-reduction attention at the config-2 shapes: student B0 (head_dim 32) forward + backward, teacher B2 (head_dim 64) forward.
    Both arithmetic modes: exact f32-input MFMA and split-bf16 (tunable sra_split_bf16)."""
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(1234)
    out = []
    for D, tag, with_bwd in ((32, 'student B0', True), (64, 'teacher B2', False)):
        for N, heads in ((16384, 1), (4096, 2), (1024, 5), (256, 8)):
            KV, C = 256, heads * D
            q = torch.randn(B, N, C, device=dev, generator=gen)
            kv = torch.randn(B, KV, 2 * C, device=dev, generator=gen)
            do = torch.randn(B, N, C, device=dev, generator=gen)
            o, dq, dkv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(kv)
            lse = torch.empty(B, heads, N, device=dev)
            wsb = L.sd_sra_workspace_bytes(B, N, KV, heads, D)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            flops = 4.0 * B * heads * N * KV * D
            for mode, mtag, peak in ((1, 'split-bf16', MFMA_BF16 / 6), (0, 'f32 MFMA', MFMA_F32)):
                _lib.set_tunable('sra_split_bf16', mode)
                tf = _time(lambda st: _ok(L.sd_sra_fwd(q.data_ptr(), kv.data_ptr(), o.data_ptr(), lse.data_ptr(), 0, B, N, KV, heads, D, D ** -0.5, st), 'sra fwd'), reps)
                note = 'peak = dense bf16 MFMA / 6 cross products' if mode else None
                out.append(_entry(f'sra fwd {tag} N={N} heads={heads} ({mtag})', 'sra_fwd_x3' if mode else 'sra_fwd', [B, N, KV, heads, D], 'f32', tf, 'mfma', flops, round(peak, 1), note))
                if with_bwd:
                    tb = _time(lambda st: _ok(L.sd_sra_bwd(q.data_ptr(), kv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), dq.data_ptr(), dkv.data_ptr(), 0, B, N, KV,
                                                           heads, D, D ** -0.5, ws.data_ptr(), wsb, st), 'sra bwd'), reps)
                    out.append(_entry(f'sra bwd {tag} N={N} heads={heads} ({mtag})', 'sra_bwd_dq + sra_bwd_dkv + sra_dkv_reduce', [B, N, KV, heads, D], 'f32', tb, 'mfma',
                                      2.5 * flops, round(peak, 1), note))
            _lib.set_tunable('sra_split_bf16', 1)
    return out


def bench_optim(dev, reps):
    """The one-launch AdamW (csrc/optim.hip) over the trainable tensors of the config-2 student (Segformer-B0 + head), next to ATen's fused path."""
    import bench
    from segdistill_amd.config import Config
    from segdistill_amd.engine.optim import HipAdamW, build_optimizer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'kd', 'cfg2_segformer_b2_b0_cgd.py'))
    model = bench.build_model(cfg, dev)
    out = []
    for env, tag in (('1', 'adamw_multi (one launch)'), ('0', 'ATen fused AdamW (multi_tensor_apply per group)')):
        os.environ['SEGDISTILL_HIP_ADAMW'] = env
        opt = build_optimizer(model, dict(cfg.optimizer))
        ps = [p for g in opt.param_groups for p in g['params']]
        gen = torch.Generator(device=dev).manual_seed(5)
        for p in ps:
            p.grad = torch.randn(p.shape, device=dev, generator=gen).contiguous(memory_format=torch.channels_last) if (p.dim() == 4 and not p.is_contiguous()) \
                else torch.randn(p.shape, device=dev, generator=gen)
        opt.step()
        assert isinstance(opt, HipAdamW) == (env == '1')
        # ours is a single launch and captures; ATen's fused step is not capturable as the trainer builds it: its time includes the host side
        t = _time(lambda st: opt.step(), reps, per_graph=4 if env == '1' else 1, graph=env == '1')
        n = sum(p.numel() for p in ps)
        out.append(_entry(f'AdamW step, {len(ps)} tensors / {n / 1e6:.2f} M parameters: {tag}', 'adamw_multi' if env == '1' else 'multi_tensor_apply x n', [len(ps), n],
                          'f32', t, 'hbm', 7 * n * 4, HBM, 'device time in a replayed graph' if env == '1' else 'eager timing: includes the host side of the step'))
    os.environ.pop('SEGDISTILL_HIP_ADAMW', None)
    return out


def bench_pix(dev, reps, B=8, C=150, HW=512):
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(1234)
    S = 2 * torch.randn(B, C, HW, HW, device=dev, generator=gen)
    T = 2 * torch.randn(B, C, HW, HW, device=dev, generator=gen)
    rows = B * HW * HW
    lse, loss, dS = torch.empty(2, rows, device=dev), torch.empty((), device=dev), torch.empty_like(S)
    wsb = L.sd_pix_kl_workspace_bytes(B, C, HW, HW)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    up = torch.ones((), device=dev)
    tf = _time(lambda st: _ok(L.sd_pix_kl_fwd(S.data_ptr(), T.data_ptr(), 0, B, C, HW, HW, 1.0, 1.0 / rows, lse.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, st),
                           'pix fwd'), reps)
    tb = _time(lambda st: _ok(L.sd_pix_kl_bwd(S.data_ptr(), T.data_ptr(), 0, B, C, HW, HW, 1.0, 1.0 / rows, lse.data_ptr(), up.data_ptr(), dS.data_ptr(), st),
                           'pix bwd'), reps)
    N = S.numel()
    return [_entry('pix_kl fwd (PDLoss at label resolution)', 'pix_fwd (+ loss)', [B, C, HW, HW], 'f32', tf, 'hbm', 2 * N * 4, HBM),
            _entry('pix_kl bwd', 'pix_bwd', [B, C, HW, HW], 'f32', tb, 'hbm', 3 * N * 4, HBM)]


def bench_at(dev, reps, B=8, C=150, hw=128):
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(1234)
    S = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    T = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    planes, loss, dS = torch.empty(3, B * hw * hw, device=dev), torch.empty((), device=dev), torch.empty_like(S)
    wsb = L.sd_pix_kl_workspace_bytes(B, C, hw, hw)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    up = torch.ones((), device=dev)
    tf = _time(lambda st: _ok(L.sd_at_kl_fwd(S.data_ptr(), T.data_ptr(), 0, B, C, hw, hw, planes.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, st), 'at fwd'), reps)
    tb = _time(lambda st: _ok(L.sd_at_kl_bwd(S.data_ptr(), T.data_ptr(), 0, B, C, hw, hw, planes.data_ptr(), up.data_ptr(), dS.data_ptr(), st), 'at bwd'), reps)
    N = S.numel()
    note = '78.6 MB operands: they fit the 256 MiB Infinity Cache when launched back to back, so the rate may exceed the HBM peak'
    return [_entry('at_kl fwd (ATLoss, logits at tap resolution)', 'at_fwd', [B, C, hw, hw], 'f32', tf, 'hbm', 2 * N * 4, HBM, note),
            _entry('at_kl bwd', 'at_bwd', [B, C, hw, hw], 'f32', tb, 'hbm', 3 * N * 4, HBM, note)]


def bench_ifvd(dev, reps, B=8, C=150, hw=128):
    """The five C-ABI calls of ops._IFVDFunction on the stream handed in (the autograd binding itself cannot be timed inside a capture here: a
    backward node runs on the stream its forward ran on).  Random labels = every class block present in every 16-pixel step: the worst case
    of the one-hot products; `blocky` = 16 x 16 patches of one class, closer to a label map."""
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(1234)
    HW, K = hw * hw, C
    S = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    T = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    f32 = dict(dtype=torch.float32, device=dev)
    counts = torch.empty(B, K, dtype=torch.int32, device=dev)
    smask = torch.empty(L.sd_ifvd_stepmask_ints(B, HW, K), dtype=torch.int32, device=dev)
    mean_s, mean_t, A, Bk = torch.empty(B, C, K, **f32), torch.empty(B, C, K, **f32), torch.empty(B, C, K, **f32), torch.empty(B, K, **f32)
    coefs, loss, dS = torch.empty(3, B * HW, **f32), torch.empty((), **f32), torch.empty_like(S)
    wsb = L.sd_ifvd_workspace_bytes(B, C, HW, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    out = []
    for tag in ('random labels', 'blocky labels'):
        if tag.startswith('random'):
            cls = torch.randint(0, C, (B, HW), device=dev, generator=gen, dtype=torch.int32)
        else:
            cls = torch.randint(0, C, (B, hw // 16, hw // 16), device=dev, generator=gen, dtype=torch.int32)
            cls = cls.repeat_interleave(16, 1).repeat_interleave(16, 2).reshape(B, HW).contiguous()

        def fwd(st):
            _ok(L.sd_ifvd_counts(cls.data_ptr(), B, HW, K, counts.data_ptr(), smask.data_ptr(), st), 'ifvd counts')
            _ok(L.sd_ifvd_class_means(S.data_ptr(), T.data_ptr(), 0, cls.data_ptr(), smask.data_ptr(), counts.data_ptr(), mean_s.data_ptr(),
                                      mean_t.data_ptr(), ws.data_ptr(), wsb, B, C, HW, K, st), 'ifvd class_means')
            _ok(L.sd_ifvd_cos(S.data_ptr(), T.data_ptr(), 0, cls.data_ptr(), mean_s.data_ptr(), mean_t.data_ptr(), coefs.data_ptr(), loss.data_ptr(),
                              ws.data_ptr(), wsb, B, C, HW, K, st), 'ifvd cos')

        def bwd(st):
            _ok(L.sd_ifvd_coef_sums(S.data_ptr(), 0, cls.data_ptr(), smask.data_ptr(), counts.data_ptr(), coefs.data_ptr(), A.data_ptr(), Bk.data_ptr(),
                                    ws.data_ptr(), wsb, B, C, HW, K, st), 'ifvd coef_sums')
            _ok(L.sd_ifvd_bwd(S.data_ptr(), 0, cls.data_ptr(), mean_s.data_ptr(), coefs.data_ptr(), A.data_ptr(), Bk.data_ptr(), counts.data_ptr(), None,
                              dS.data_ptr(), B, C, HW, K, st), 'ifvd bwd')

        tf = _time(fwd, reps)
        tb = _time(bwd, reps)
        N = S.numel()
        note = 'L2/Infinity-Cache resident at this size; round 2 (torch sort chain + one gather launch per network): 0.72 / 0.40 ms eager, 0.65 ms device'
        out += [_entry(f'ifvd fwd, {tag} (counts + both networks\' class means + cosine pass + loss)',
                       'ifvd_counts + ifvd_onehot_sums + ifvd_finish + ifvd_cos + ifvd_loss', [B, C, hw, hw], 'f32', tf, 'hbm', 4 * N * 4, HBM, note),
                _entry(f'ifvd bwd, {tag}', 'ifvd_onehot_sums (weighted) + ifvd_finish + ifvd_bwd', [B, C, hw, hw], 'f32', tb, 'hbm', 3 * N * 4, HBM, note)]
    return out


def bench_ce(dev, reps, B=8, C=150, hw=128, F=4):
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(1234)
    x = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    H = hw * F
    lab = torch.randint(0, C, (B, H, H), device=dev, generator=gen, dtype=torch.int32)
    lab[torch.rand(B, H, H, device=dev, generator=gen) < 0.05] = 255
    loss_pix, lse2 = torch.empty(B, H, H, device=dev), torch.empty(B, H, H, device=dev)
    correct = torch.empty(1, dtype=torch.int32, device=dev)
    dx = torch.empty_like(x)
    up = torch.full((1,), 1.0 / (B * H * H), device=dev)
    tf = _time(lambda st: _ok(L.sd_ce_up_fwd(x.data_ptr(), lab.data_ptr(), loss_pix.data_ptr(), lse2.data_ptr(), correct.data_ptr(), 0, B, C, hw, hw, H, H, 255, st),
                           'ce fwd'), reps)
    tb = _time(lambda st: _ok(L.sd_ce_up_bwd(x.data_ptr(), lab.data_ptr(), lse2.data_ptr(), up.data_ptr(), 0, 1.0, dx.data_ptr(), 0, B, C, hw, hw, H, H, 255, st),
                           'ce bwd'), reps)
    N = B * C * H * H
    note = 'exp/VALU-bound (reads only the taps + the label map); lane-instruction MODEL: ~9 VALU + 1 exp (x4) per output pixel and class'
    return [_entry('ce_up fwd (fused x4 upsample + log-softmax + NLL + top-1)', 'ce_up_fwd_col', [B, C, hw, hw, '->', H, H], 'f32', tf, 'valu', N * (9 + 4), VALU, note),
            _entry('ce_up bwd', 'ce_up_bwd', [B, C, hw, hw, '->', H, H], 'f32', tb, 'valu', N * (8 + 4), VALU, note)]


def bench_dw(dev, reps, B=8):
    """Depth-wise 3x3 (+ bias + erf GELU) of the Mix-FFN, token-major [B, H*W, 4*dim]: the frozen teacher's stage-1 map of config 2 and the
    student's training forward / input gradient / weight-gradient partials at its stage-1 map."""
    from segdistill_amd import _lib
    L = _lib.lib()
    out = []
    for tag, side, C, train in (('teacher stage 1 (frozen)', 128, 256, False), ('student stage 1 (training)', 128, 128, True)):
        x = torch.randn(B, side * side, C, device=dev)
        w = torch.randn(C, 9, device=dev) / 3
        b = torch.randn(C, device=dev)
        y, pre, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        nb = x.numel() * 4
        shape = [B, side, side, C]
        if not train:
            t = _time(lambda st: _ok(L.sd_dwconv3x3_gelu_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 0, B, side, side, C, st), 'dw gelu'), reps)
            out.append(_entry(f'dw3x3 + GELU fwd, {tag}', 'dw3x3_fwd', shape, 'f32', t, 'hbm', 2 * nb, HBM))
            continue
        t = _time(lambda st: _ok(L.sd_dwconv3x3_gelu_fwd_train(x.data_ptr(), w.data_ptr(), b.data_ptr(), pre.data_ptr(), y.data_ptr(), 0, B, side, side, C, st),
                                 'dw gelu train'), reps)
        out.append(_entry(f'dw3x3 + GELU fwd keeping the pre-activation, {tag}', 'dw3x3_fwd', shape, 'f32', t, 'hbm', 3 * nb, HBM))
        t = _time(lambda st: _ok(L.sd_dwconv3x3_bwd_data(y.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, B, side, side, C, st), 'dw bwd data'), reps)
        out.append(_entry(f'dw3x3 bwd_data, {tag}', 'dw3x3_fwd<flip>', shape, 'f32', t, 'hbm', 2 * nb, HBM))
        wsb = L.sd_dwconv3x3_workspace_bytes(0, B, side, side, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        t = _time(lambda st: _ok(L.sd_dwconv3x3_bwd_weight(x.data_ptr(), y.data_ptr(), None, None, 0, B, side, side, C, ws.data_ptr(), wsb, st), 'dw bwd weight'), reps)
        out.append(_entry(f'dw3x3 bwd_weight partials (combine deferred), {tag}', 'dw3x3_wgrad_partials', shape, 'f32', t, 'hbm', 2 * nb, HBM))
    return out


def bench_ln(dev, reps, rows=131072, C=64):
    """LayerNorm over the channels of token-major activations at the teacher's stage-1 shape of config 2."""
    from segdistill_amd import _lib
    L = _lib.lib()
    x, dy = torch.randn(rows, C, device=dev), torch.randn(rows, C, device=dev)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    wsb = L.sd_layernorm_workspace_bytes(rows, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    nb = x.numel() * 4
    tf = _time(lambda st: _ok(L.sd_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0, rows, C, 1e-6, st),
                              'ln fwd'), reps)
    tb = _time(lambda st: _ok(L.sd_layernorm_bwd(x.data_ptr(), dy.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), None, None, 0, rows, C,
                                                 ws.data_ptr(), wsb, st), 'ln bwd'), reps)
    return [_entry('layernorm fwd', 'ln_fwd', [rows, C], 'f32', tf, 'hbm', 2 * nb, HBM),
            _entry('layernorm bwd (parameter-gradient combine deferred)', 'ln_bwd', [rows, C], 'f32', tb, 'hbm', 3 * nb, HBM)]


def bench_upsum(dev, reps, B=8, E=256):
    """y = z1 + up(z2) + up(z3) + up(z4) of the SegFormer head (student, E = 256) and its backward to the three coarse branches."""
    from segdistill_amd import _lib
    L = _lib.lib()
    sizes = [(128, 128), (64, 64), (32, 32), (16, 16)]
    zs = [torch.randn(B, h * w, E, device=dev) for h, w in sizes]
    y = torch.empty_like(zs[0])
    dz = [torch.empty_like(z) for z in zs[1:]]
    nb = [z.numel() * 4 for z in zs]
    tf = _time(lambda st: _ok(L.sd_upsum_fwd(zs[0].data_ptr(), zs[1].data_ptr(), zs[2].data_ptr(), zs[3].data_ptr(), None, y.data_ptr(), 0, B, 128, 128, E, 2, 4, 8, st),
                              'upsum fwd'), reps)
    wsb = L.sd_upsum_bwd3_workspace_bytes(B, 128, 128, E)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    tb = _time(lambda st: _ok(L.sd_upsum_bwd3(y.data_ptr(), dz[0].data_ptr(), dz[1].data_ptr(), dz[2].data_ptr(), 0, B, 128, 128, E, ws.data_ptr(), wsb, st), 'upsum bwd3'), reps)
    return [_entry('head up-sample + sum fwd', 'upsum_fwd_strip', [B, 128, 128, E], 'f32', tf, 'hbm', 2 * nb[0] + sum(nb[1:]), HBM),
            _entry('head up-sample + sum bwd (three coarse branches, separable)', 'upsum_bwd_rows + upsum_bwd_cols', [B, 128, 128, E], 'f32', tb, 'hbm',
                   nb[0] + sum(nb[1:]), HBM, 'algorithmic bytes: dy once + the three gradients; the fp32 row partials (7/8 of dy, written and re-read) are extra traffic')]


def bench_resize(dev, reps):
    """csrc/resize.hip at the shapes of the config-4 teacher's UPerHead (level fusion [8,512,64,64] -> 128^2) and of the PSPNet student's
    pooled branches ([8,128,6,6] -> 64^2), forward and gather backward."""
    from segdistill_amd import _lib
    L = _lib.lib()
    out = []
    for (B, C, h, H, tag) in ((8, 512, 64, 128, 'UPerHead level fusion x2'), (8, 512, 32, 128, 'UPerHead level x4'), (8, 512, 16, 128, 'UPerHead coarsest level x8'),
                              (8, 128, 6, 64, 'PPM branch 6x6 -> 64x64')):
        x = torch.randn(B, C, h, h, device=dev)
        y = torch.empty(B, C, H, H, device=dev)
        dx = torch.empty_like(x)
        tf = _time(lambda st: _ok(L.sd_resize_bilinear_fwd(x.data_ptr(), y.data_ptr(), 0, B * C, h, h, H, H, 0, st), 'resize fwd'), reps)
        wsb = L.sd_resize_bilinear_bwd_workspace_bytes(B * C, h, H)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        tb = _time(lambda st: _ok(L.sd_resize_bilinear_bwd(y.data_ptr(), dx.data_ptr(), 0, B * C, h, h, H, H, 0, ws.data_ptr(), wsb, st), 'resize bwd'), reps)
        nb_in, nb_out = x.numel() * 4, y.numel() * 4
        out += [_entry(f'bilinear resize fwd ({tag})', 'resize_bilinear_fwd', [B, C, h, h, H, H], 'f32', tf, 'hbm', nb_in + nb_out, HBM),
                _entry(f'bilinear resize bwd ({tag})', 'resize_bilinear_bwd', [B, C, h, h, H, H], 'f32', tb, 'hbm', nb_in + nb_out, HBM,
                       'integer factors: one kernel, x-fold from global, y-fold through LDS; other sizes: separable gather through an fp32 workspace')]
    return out


def bench_gemm_planes(dev, reps):
    """Token Linear products on pre-split weight planes (sd_linear_fwd_planes / _bwd_data_planes) at the largest shapes of config 2, priced
    against the split-bf16 matrix-pipe bound (dense bf16 / 6 cross products) and, for the HBM-bound ones, against 8 TB/s."""
    from segdistill_amd import planes, token_gemm
    out = []
    for (T, K, N, tag) in ((131072, 256, 256, 'head fuse1 256->256'), (131072, 64, 256, 'teacher s1 fc1 64->256'), (32768, 128, 512, 'teacher s2 fc1'),
                           (8192, 320, 1280, 'teacher s3 fc1'), (8192, 1280, 320, 'teacher s3 fc2')):
        x = torch.randn(T, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        pf = planes.get(w, 'fwd')
        t = _time(lambda st: token_gemm.linear_fwd_planes(x, w, pf, b), reps)
        flops, nbytes = 2.0 * T * K * N, (T * K + T * N) * 4.0
        if flops / (MFMA_BF16 / 6 * 1e12) >= nbytes / (HBM * 1e9):
            out.append(_entry(f'token Linear fwd on weight planes ({tag})', 'token_gemm_f32<..., BFRAG>', [T, K, N], 'f32 (split-bf16)', t, 'mfma', flops,
                              MFMA_BF16 / 6, 'bound: dense bf16 peak / 6 cross products'))
        else:
            out.append(_entry(f'token Linear fwd on weight planes ({tag})', 'token_gemm_f32<..., BFRAG>', [T, K, N], 'f32 (split-bf16)', t, 'hbm', nbytes, HBM))
    return out


def bench_pred(dev, reps):
    """linear_pred as class planes (sd_linear_nchw_fwd_planes: W as pre-split row planes through LDS) for the teacher (E = 768) and the student."""
    from segdistill_amd import _lib, planes
    L = _lib.lib()
    out = []
    for E in (768, 256):
        B, P, N = 8, 16384, 150
        x = torch.randn(B, P, E, device=dev)
        w = torch.randn(N, E, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        y = torch.empty(B, N, P, device=dev)
        pr = planes.get(w, 'rows')
        t = _time(lambda st: _ok(L.sd_linear_nchw_fwd_planes(x.data_ptr(), pr.data_ptr(), b.data_ptr(), y.data_ptr(), 0, B, P, E, N, st), 'pred'), reps)
        out.append(_entry(f'linear_pred -> class planes, E = {E}', 'token_gemm_f32<160, 128, 1, 4, ..., APL>', [B, P, E, N], 'f32 (split-bf16)', t, 'mfma',
                          2.0 * B * P * E * N, MFMA_BF16 / 6, 'bound: dense bf16 peak / 6 cross products'))
    return out


def bench_wgrad_bf16(dev, reps):
    """dW = dY^T . X of token-major Linears under bf16 storage (config 5: the 256 -> 768 align projections of the four decoder stages and the
    B1 student's Mix-FFN weights): sd_linear_wgrad_generic_partials = csrc/wgrad_tn.hip since round 4 (slabs only; the combine is deferred)."""
    from segdistill_amd import _lib
    L = _lib.lib()
    out = []
    for tag, T, M, N in (('align s1 768x256 over 131072', 131072, 768, 256), ('align s2 768x256 over 32768', 32768, 768, 256),
                         ('align s3 768x256 over 8192', 8192, 768, 256), ('B1 s3 fc1 1280x320 over 8192', 8192, 1280, 320),
                         ('B1 s4 fc1 2048x512 over 2048', 2048, 2048, 512), ('B1 s2 fc2 128x512 over 32768', 32768, 128, 512)):
        gen = torch.Generator(device=dev).manual_seed(3)
        dy = torch.randn(T, M, device=dev, generator=gen).to(torch.bfloat16)
        x = torch.randn(T, N, device=dev, generator=gen).to(torch.bfloat16)
        ns = L.sd_linear_wgrad_generic_slabs(1, T, M, N)
        if ns == 0:
            continue
        ws = torch.empty(ns * M * N, dtype=torch.float32, device=dev)
        t = _time(lambda st: _ok(L.sd_linear_wgrad_generic_partials(dy.data_ptr(), x.data_ptr(), 1, T, M, N, ws.data_ptr(), ws.numel() * 4, st), 'wgrad'), reps)
        nbytes = 2.0 * T * (M + N) + 4.0 * ns * M * N
        out.append(_entry(f'bf16 Linear weight gradient, {tag} ({ns} slabs)', 'wgrad_tn_bf16', [T, M, N], 'bf16', t, 'hbm', nbytes, HBM,
                          'bytes = dY + X once + the fp32 slabs written'))
    return out


def bench_wgrad_multi(dev, reps):
    """The grouped weight-gradient launches (round 5; csrc/wgrad_tn.hip wgrad_tn_x3_multi / wgrad_tn_bf16_ring_multi + multi_slab_reduce): every
    token-major Linear of the student of config 2 (Segformer-B0, fp32, bias sums in the slabs) resp. config 5 (B1, bf16) in one deferred scope.
    Roof: HBM -- every dY and X once + the slabs written and read back."""
    import ctypes as C_
    from segdistill_amd import _lib, deferred
    from tools.slab_budget import PRESETS
    L = _lib.lib()
    out = []
    for tag, preset, dt, code in (('config 2 (B0 student, fp32, bias sums riding along)', 'cfg2', torch.float32, 0),
                                  ('config 5 (B1 student + align, bf16)', 'cfg5', torch.bfloat16, 1)):
        pr = PRESETS[preset]
        dims, embed, align = pr['dims'], pr['embed'], pr['align']
        shapes = []
        for s_, c in enumerate(dims):
            t = 8 * (512 // (4 << s_)) ** 2
            tk = t // (8, 4, 2, 1)[s_] ** 2
            layers = [(t, c, c), (tk, 2 * c, c), (t, c, c), (t, 4 * c, c), (t, c, 4 * c)]
            shapes += layers * 2 + [(t, embed, c)] + ([(t, align, embed)] if align else [])
        shapes = [(T, M, N) for T, M, N in shapes if L.sd_linear_wgrad_tn_multi_supported(code, T, M, N)]
        gen = torch.Generator(device=dev).manual_seed(17)
        ops_ = [(torch.randn(T, M, device=dev, generator=gen).to(dt), torch.randn(T, N, device=dev, generator=gen).to(dt)) for T, M, N in shapes]
        wb = dt == torch.float32
        arr = (deferred._WgradJob * len(shapes))()
        for k, ((dy, x), (T, M, N)) in enumerate(zip(ops_, shapes)):
            arr[k].dY, arr[k].X, arr[k].tokens, arr[k].out_features, arr[k].in_features, arr[k].with_bias = dy.data_ptr(), x.data_ptr(), T, M, N, int(wb)
        _ok(L.sd_linear_wgrad_tn_multi_plan(C_.cast(arr, C_.c_void_p), len(shapes), code), 'plan')
        sizes = [M * N + (M if wb else 0) for T, M, N in shapes]
        outs = [torch.empty(n, device=dev) for n in sizes]
        ws = torch.empty(sum(arr[k].nsplit * sizes[k] for k in range(len(shapes)) if arr[k].nsplit > 1) + 4, device=dev)
        red, off = [], 0
        for k in range(len(shapes)):
            if arr[k].nsplit > 1:
                arr[k].slabs = ws[off:].data_ptr()
                red.append((ws[off:].data_ptr(), outs[k].data_ptr(), sizes[k], arr[k].nsplit))
                off += arr[k].nsplit * sizes[k]
            else:
                arr[k].slabs = outs[k].data_ptr()
        jobs = (deferred._Job * len(red))()
        for k, (p_, o_, n_, ns_) in enumerate(red):
            jobs[k].partials, jobs[k].out, jobs[k].n, jobs[k].nslabs = p_, o_, n_, ns_

        def run(st):
            _ok(L.sd_linear_wgrad_tn_multi(C_.cast(arr, C_.c_void_p), len(shapes), code, st), 'wgrad multi')
            _ok(L.sd_multi_slab_reduce(C_.cast(jobs, C_.c_void_p), len(red), st), 'slab reduce')
        t = _time(run, reps)
        es = 4 if wb else 2
        slab = sum(arr[k].nsplit * sizes[k] * 4 for k in range(len(shapes)) if arr[k].nsplit > 1)
        nbytes = sum(float(T) * (M + N) * es for T, M, N in shapes) + 2.0 * slab
        out.append(_entry(f'all {len(shapes)} Linear weight gradients of one backward, grouped: {tag}', 'wgrad_tn_*_multi + multi_slab_reduce',
                          [len(shapes)], 'f32' if wb else 'bf16', t, 'hbm', nbytes, HBM, note=f'{slab / 1e6:.0f} MB of slabs written + read back'))
    return out


def bench_ppm(dev, reps):
    """All adaptive average pools of a PPM in one pass each way (csrc/ppm_pool.hip): the PSPNet-R18 student's [8, 512, 64, 64] map and the
    ResNet-101 teacher's [2, 2048, 64, 64] (config 1).  HBM-bound: forward reads the map once, backward writes it once."""
    import ctypes as C_
    from segdistill_amd import _lib
    L = _lib.lib()
    out = []
    scales = (1, 2, 3, 6)
    sc = (C_.c_int * 4)(*scales)
    for tag, B, C, h in (('PSPNet-R18 student [8,512,64,64]', 8, 512, 64), ('PSPNet-R101 [2,2048,64,64]', 2, 2048, 64)):
        x = torch.randn(B, C, h, h, device=dev)
        pooled = [torch.empty(B, C, s, s, device=dev) for s in scales]
        dx = torch.empty_like(x)
        ptrs = (C_.c_void_p * 4)(*[p.data_ptr() for p in pooled])
        tf = _time(lambda st: _ok(L.sd_ppm_pool_fwd(x.data_ptr(), 0, B * C, h, h, sc, 4, ptrs, st), 'ppm fwd'), reps)
        tb = _time(lambda st: _ok(L.sd_ppm_pool_bwd(ptrs, 0, B * C, h, h, sc, 4, dx.data_ptr(), st), 'ppm bwd'), reps)
        nbytes = x.numel() * 4
        out += [_entry(f'PPM pools (1, 2, 3, 6) fwd, {tag}', 'ppm_pool_fwd', [B, C, h, h], 'f32', tf, 'hbm', nbytes, HBM),
                _entry(f'PPM pools (1, 2, 3, 6) bwd, {tag}', 'ppm_pool_bwd', [B, C, h, h], 'f32', tb, 'hbm', nbytes, HBM)]
    return out


def bench_wattn(dev, reps):
    """Window attention of the frozen Swin-B teacher (config 4: 512 x 512 crops, batch 8, 7 x 7 windows) -- csrc/window_attn.hip, one launch per
    block over the qkv Linear's output, f32 MFMA.  Bytes = qkv read once + the token-major output written (bias / mask tables are L2-resident).
    The matrix work (128 MFMAs of 64 cycles per window and head) is ~46 us at stage 1 -- the same order as the 53 us the bytes take at 5.5 TB/s."""
    from segdistill_amd import _lib
    L = _lib.lib()
    out = []
    for tag, windows, nw, heads in (('stage 1: 2888 windows x 4 heads', 2888, 361, 4), ('stage 2: 800 windows x 8 heads', 800, 100, 8),
                                    ('stage 3: 200 windows x 16 heads', 200, 25, 16), ('stage 4: 72 windows x 32 heads', 72, 9, 32)):
        C = heads * 32
        gen = torch.Generator(device=dev).manual_seed(5)
        qkv = torch.randn(windows, 49, 3 * C, device=dev, generator=gen)
        bias = torch.randn(heads, 49, 49, device=dev, generator=gen)
        mask = torch.where(torch.rand(nw, 49, 49, device=dev, generator=gen) < 0.3, -100.0, 0.0)
        o = torch.empty(windows, 49, C, device=dev)
        nbytes = 4.0 * windows * 49 * C * 4
        from segdistill_amd import window_attn
        bias_p = window_attn.pack_tables(bias, float('-inf'))[0]
        mask[::3] = 0.0                                                      # some windows without a mask, as in a shifted partition
        mask_p, flags = window_attn.pack_tables(mask, 0.0, True)
        for shifted in (False, True):
            t = _time(lambda st: _ok(L.sd_window_attn_fwd_packed(qkv.data_ptr(), bias_p.data_ptr(), mask_p.data_ptr() if shifted else None,
                                                                 flags.data_ptr() if shifted else None, o.data_ptr(), 0, windows,
                                                                 nw if shifted else 0, heads, 49, 32, 32 ** -0.5, st), 'wattn mfma'), reps)
            out.append(_entry(f'Swin-B window attention fwd {tag}{" (shifted: mask)" if shifted else ""}', 'window_attn_mfma',
                              [windows, 49, 3 * C], 'f32', t, 'hbm', nbytes, HBM, 'bytes = qkv + out once'))
    return out


def bench_aligntok(dev, reps, B=8, K=256, C=768, g=8, stages=(16384, 4096, 1024, 256)):
    """csrc/align_tok.hip at config 5's shapes: the fused projection + criterion forward / backward (stage 1 alone and all four stages in one
    call), the stand-alone projection and the input-gradient GEMM.  Roof: HBM (X + T forward, X + T + dY backward, X + Y / dY + dX for the
    GEMMs); the matrix-pipe view (2 T K C flops against dense bf16) is in the note."""
    import ctypes as C_
    from segdistill_amd import _lib, ops
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(1234)
    xs = [torch.randn(B, P, K, device=dev, generator=gen).to(torch.bfloat16) for P in stages]
    ts = [(2 * torch.randn(B, P, C, device=dev, generator=gen)).to(torch.bfloat16) for P in stages]
    w = (torch.randn(C, K, device=dev, generator=gen) / K ** 0.5).to(torch.bfloat16)
    bias = 0.1 * torch.randn(C, device=dev, generator=gen)
    rows = B * (-(-C // g))
    keep = []

    def jobs(idx):
        arr = (ops._AlignTokJob * len(idx))()
        for k, i in enumerate(idx):
            P = stages[i]
            wsb = L.sd_align_cgd_tok_workspace_bytes(B, C, P)
            bufs = [torch.empty(wsb, dtype=torch.uint8, device=dev), torch.zeros(rows, 2, device=dev), torch.empty(rows, device=dev), torch.empty((), device=dev),
                    torch.empty(B, P, C, dtype=torch.bfloat16, device=dev), torch.empty(L.sd_align_cgd_tok_tiles(B, P), C, device=dev)]
            keep.extend(bufs)
            j = arr[k]
            j.X, j.W, j.bias, j.T = xs[i].data_ptr(), w.data_ptr(), bias.data_ptr(), ts[i].data_ptr()
            j.workspace, j.workspace_bytes, j.row_lse2, j.row_kl, j.loss = bufs[0].data_ptr(), wsb, bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr()
            j.out, j.db_part = bufs[4].data_ptr(), bufs[5].data_ptr()
            j.P, j.B, j.K, j.C, j.g, j.inv_tau, j.loss_scale, j.coef = P, B, K, C, g, 0.25, 3.0 / rows, 3.0 / (rows * 4.0)
        return arr

    out = []
    for idx, tag in (([0], 'cfg5 stage 1'), (list(range(len(stages))), 'cfg5 all four stages in one call')):
        arr = jobs(idx)
        n = len(idx)
        ptr = C_.cast(arr, C_.c_void_p)
        _ok(L.sd_align_cgd_tok_fwd_multi(ptr, n, None), 'align_tok fwd')       # row constants for the backward
        tf = _time(lambda st: _ok(L.sd_align_cgd_tok_fwd_multi(ptr, n, st), 'align_tok fwd'), reps)
        tb = _time(lambda st: _ok(L.sd_align_cgd_tok_bwd_multi(ptr, n, st), 'align_tok bwd'), reps)
        T = sum(B * stages[i] for i in idx)
        flops = 2.0 * T * K * C
        shape = [[B, stages[i], K, C] for i in idx]
        out.append(_entry(f'align + criterion fused fwd, {tag} (bf16)', 'align_tok_kernel<16,0> + cgd_tok_finish', shape, 'bf16', tf, 'hbm', T * (K + C) * 2, HBM,
                          note=f'{flops / (tf * 1e-3) / 1e12:.0f} TFLOP/s of {MFMA_BF16:.0f} (bf16 MFMA); Y never written'))
        out.append(_entry(f'align + criterion fused bwd (recompute, dY, db), {tag} (bf16)', 'align_tok_kernel<16,1>', shape, 'bf16', tb, 'hbm', T * (K + 2 * C) * 2, HBM,
                          note=f'{flops / (tb * 1e-3) / 1e12:.0f} TFLOP/s of {MFMA_BF16:.0f}'))
    T0 = B * stages[0]
    y = torch.empty(T0, C, dtype=torch.bfloat16, device=dev)
    dx = torch.empty(T0, K, dtype=torch.bfloat16, device=dev)
    x2 = xs[0].view(T0, K)
    tp = _time(lambda st: _ok(L.sd_linear_tok_bf16_fwd(x2.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), T0, K, C, st), 'plain'), reps)
    td = _time(lambda st: _ok(L.sd_linear_tok_bf16_bwd_data(y.data_ptr(), w.data_ptr(), dx.data_ptr(), T0, C, K, st), 'dx'), reps)
    gemm_bytes, gemm_flops = T0 * (K + C) * 2, 2.0 * T0 * K * C
    out.append(_entry('align projection alone Y = X.W^T + b, cfg5 stage 1 (bf16)', 'align_tok_kernel<16,2>', [T0, K, C], 'bf16', tp, 'hbm', gemm_bytes, HBM,
                      note=f'{gemm_flops / (tp * 1e-3) / 1e12:.0f} TFLOP/s of {MFMA_BF16:.0f}'))
    out.append(_entry('align input gradient dX = dY.W, cfg5 stage 1 (bf16)', 'tok_dx_kernel<8>', [T0, C, K], 'bf16', td, 'hbm', gemm_bytes, HBM,
                      note=f'{gemm_flops / (td * 1e-3) / 1e12:.0f} TFLOP/s of {MFMA_BF16:.0f}'))
    return out


def bench_pixup(dev, reps, B=8, C=150, hw=128, F=4):
    """csrc/pix_up.hip: PDLoss with the up-sampling fused, config-2 taps.  VALU-bound like R2: priced by lane-instructions (19 per interpolated
    (s, t) pair forward, 12 backward)."""
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(1234)
    s = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    t = 2 * torch.randn(B, C, hw, hw, device=dev, generator=gen)
    H = W = F * hw
    rows = B * H * W
    lse, loss, ds = torch.empty(2, rows, device=dev), torch.empty((), device=dev), torch.empty_like(s)
    wsb = L.sd_pix_kl_up_workspace_bytes(B, hw)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    up = torch.ones((), device=dev)
    tf = _time(lambda st: _ok(L.sd_pix_kl_up_fwd(s.data_ptr(), t.data_ptr(), 0, B, C, hw, hw, H, W, 1.0, 1.0 / rows, lse.data_ptr(), loss.data_ptr(),
                                                  ws.data_ptr(), wsb, st), 'pix_up fwd'), reps)
    tb = _time(lambda st: _ok(L.sd_pix_kl_up_bwd(s.data_ptr(), t.data_ptr(), 0, B, C, hw, hw, H, W, 1.0, 1.0 / rows, lse.data_ptr(), up.data_ptr(),
                                                  ds.data_ptr(), st), 'pix_up bwd'), reps)
    N = B * C * H * W
    return [_entry('pix_kl with the x4 upsample fused, fwd (PDLoss from the taps)', 'pix_up_fwd + pix_up_loss', [B, C, hw, hw, '->', H, W], 'f32', tf, 'valu', 19 * N, VALU),
            _entry('pix_kl with the x4 upsample fused, bwd', 'pix_up_bwd', [B, C, hw, hw, '->', H, W], 'f32', tb, 'valu', 12 * N, VALU)]


def bench_bf16gemm(dev, reps):
    """csrc/tok_gemm_bf16.hip at the Segformer-B4 teacher's stage-3 / stage-2 / stage-1 shapes (config 5).  Roof: the larger of the matrix-pipe time
    (2 T K N flops on dense bf16) and the HBM time of X + W + Y -- a few microseconds for all of them: these launches are latency-bound, the
    fraction says how far above their roof one launch sits."""
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=dev).manual_seed(99)
    out = []
    for T, K, N, tag in ((8192, 320, 320, 'B4 stage 3 q / proj'), (8192, 320, 1280, 'B4 stage 3 fc1'), (8192, 1280, 320, 'B4 stage 3 fc2'),
                         (2048, 320, 640, 'B4 stage 3 kv'), (32768, 128, 128, 'B4 stage 2 q / proj'), (32768, 512, 128, 'B4 stage 2 fc2'),
                         (131072, 64, 64, 'B4 stage 1 q / proj'), (131072, 256, 64, 'B4 stage 1 fc2')):
        x = torch.randn(T, K, device=dev, generator=gen).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=gen) / K ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev, generator=gen).to(torch.bfloat16)
        y = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        t = _time(lambda st: _ok(L.sd_linear_bf16_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), _lib.SD_BF16, y.data_ptr(), T, K, N, st), 'sd_linear_bf16_fwd'), reps)
        flops, nbytes = 2.0 * T * K * N, (T * K + N * K + T * N) * 2
        if flops / (MFMA_BF16 * 1e12) > nbytes / (HBM * 1e9):
            out.append(_entry(f'bf16 Linear fwd {T} x {K} -> {N} ({tag})', 'tok_gemm_bf16_kernel', [T, K, N], 'bf16', t, 'mfma', flops, MFMA_BF16,
                              note=f'{nbytes / (t * 1e-3) / 1e9:.0f} GB/s of operands'))
        else:
            out.append(_entry(f'bf16 Linear fwd {T} x {K} -> {N} ({tag})', 'tok_gemm_bf16_kernel', [T, K, N], 'bf16', t, 'hbm', nbytes, HBM,
                              note=f'{flops / (t * 1e-3) / 1e12:.0f} TFLOP/s of {MFMA_BF16:.0f}'))
    return out


def bench_tails(dev, reps, B=8):
    """The frozen teacher's two tail kernels (round 6): Mix-FFN tail at the stage-1 / stage-2 maps of Segformer-B2 (fp32) and B4 (bf16), graded on HBM
    (h in + y out: what a one-pass kernel has to move); the SegFormer head tail at E = 768, graded on the matrix pipe (160 class rows, 6 products)."""
    from segdistill_amd import _lib, planes
    L = _lib.lib()
    out = []
    for side, dim in ((128, 64), (64, 128)):
        for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
            C = 4 * dim
            h = torch.randn(B, side * side, C, device=dev).to(dt)
            w = torch.randn(C, 9, device=dev) / 3
            b = torch.randn(C, device=dev)
            w2 = torch.randn(dim, C, device=dev) / C ** 0.5
            b2 = torch.randn(dim, device=dev)
            y = torch.empty(B, side * side, dim, device=dev, dtype=dt)
            t = _time(lambda st: _ok(L.sd_mixffn_tail(h.data_ptr(), w.data_ptr(), b.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), code, B, side, side,
                                                      C, dim, st), 'mixffn tail'), reps)
            nb = (h.numel() + y.numel()) * h.element_size()
            out.append(_entry(f'Mix-FFN tail (dw3x3 + GELU + fc2), {side}x{side}, {C} -> {dim}, {"f32" if code == 0 else "bf16"}', 'mixffn_tail_x3',
                              [B, side, side, C, dim], 'f32 (split-bf16)' if code == 0 else 'bf16', t, 'hbm', nb, HBM,
                              'VALU-paced (GELU): profiles/r06_mixffn_tail_stamps.txt'))
    side, E, N = 128, 768, 150
    sizes = [(side, side), (side // 2, side // 2), (side // 4, side // 4), (side // 8, side // 8)]
    zs = [torch.randn(B, hh * ww, E, device=dev) for (hh, ww) in sizes]
    scale, shift = torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev)
    wp = torch.randn(N, E, device=dev) / E ** 0.5
    bp = torch.randn(N, device=dev)
    pr = planes.get(wp, 'rows')
    lo = torch.empty(B, N, side, side, device=dev)
    t = _time(lambda st: _ok(L.sd_head_tail_f32(zs[0].data_ptr(), zs[1].data_ptr(), zs[2].data_ptr(), zs[3].data_ptr(), None, scale.data_ptr(),
                                                shift.data_ptr(), pr.data_ptr(), bp.data_ptr(), lo.data_ptr(), B, side, side, E, N, st), 'head tail'), reps)
    e = _entry(f'SegFormer head tail (sum + norm + ReLU + linear_pred), E = {E}', 'head_tail_x3', [B, side, side, E, N], 'f32 (split-bf16)', t, 'mfma',
               2.0 * B * side * side * E * N, MFMA_BF16 / 6, 'bound: dense bf16 peak / 6 cross products; profiles/r06_head_tail_stamps.txt')
    nb = (zs[0].numel() + lo.numel()) * 4
    e['hbm_GBps'] = round(nb / (t * 1e-3) / 1e9, 1)
    e['hbm_frac'] = round(nb / (t * 1e-3) / 1e9 / HBM, 4)
    out.append(e)
    return out


GROUPS = {
    'tails': lambda dev, reps: bench_tails(dev, reps),
    'wgradmulti': lambda dev, reps: bench_wgrad_multi(dev, reps),
    'bf16gemm': lambda dev, reps: bench_bf16gemm(dev, reps),
    'aligntok': lambda dev, reps: bench_aligntok(dev, reps),
    'pixup': lambda dev, reps: bench_pixup(dev, reps),
    'wattn': lambda dev, reps: bench_wattn(dev, reps),
    'ppm': lambda dev, reps: bench_ppm(dev, reps),
    'wgrad_bf16': lambda dev, reps: bench_wgrad_bf16(dev, reps),
    'r1': lambda dev, reps: bench_r1(dev, reps),
    'r1_bf16': lambda dev, reps: bench_r1(dev, reps, C=768, HW=128, dtype=torch.bfloat16),      # config 5 stage 1
    'r2': lambda dev, reps: bench_r2(dev, reps),
    'tok': lambda dev, reps: bench_tok(dev, reps),
    # config 4 (fp32, NCHW taps) is THE user of the NCHW align kernels; config 5's token-major taps take csrc/align_tok.hip (group 'aligntok');
    # the bf16 NCHW entry is kept as the direct-caller reference of the C ABI
    'align': lambda dev, reps: (bench_align(dev, reps, 8, 128, 512, 64, torch.float32, 'cfg4 f32')
                                + bench_align(dev, reps, 8, 256, 768, 128, torch.float32, 'C 256->768 at 128x128 f32')
                                + bench_align(dev, reps, 8, 256, 768, 128, torch.bfloat16, 'C 256->768 at 128x128 bf16 (NCHW, direct callers)')),
    'pix': lambda dev, reps: bench_pix(dev, reps),
    'at': lambda dev, reps: bench_at(dev, reps),
    'sra': lambda dev, reps: bench_sra(dev, reps),
    'optim': lambda dev, reps: bench_optim(dev, reps),
    'ifvd': lambda dev, reps: bench_ifvd(dev, reps),
    'ce': lambda dev, reps: bench_ce(dev, reps),
    'dw': lambda dev, reps: bench_dw(dev, reps),
    'ln': lambda dev, reps: bench_ln(dev, reps),
    'upsum': lambda dev, reps: bench_upsum(dev, reps),
    'resize': lambda dev, reps: bench_resize(dev, reps),
    'gemm': lambda dev, reps: bench_gemm_planes(dev, reps),
    'pred': lambda dev, reps: bench_pred(dev, reps),
}


def run(device, only=None, reps=20):
    out = []
    for name, fn in GROUPS.items():
        if only and name not in only:
            continue
        print(f'[kernel_rooflines] {name} ...', file=sys.stderr, flush=True)
        out += fn(device, reps)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    res = run(dev, a.only.split(',') if a.only else None, a.reps)
    for e in res:
        extra = f"  [{e['hbm_GBps']} GB/s = {e['hbm_frac']:.2%} of HBM]" if 'hbm_GBps' in e else ''
        print(f"{e['name']:<58} {e['ms']:8.4f} ms  {e['achieved']:10.2f} {e['unit']:<14} {e['frac']:7.2%} of {e['bound']} peak{extra}")
    if a.json:
        json.dump(res, open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()

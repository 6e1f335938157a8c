#!/usr/bin/env python3
"""bench.py -- KD train_step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full KD training iteration on one synthetic batch per rank:
student fwd + frozen-teacher fwd + distillation loss (HIP kernels) + supervised CE + backward +
flat-buffer gradient all-reduce (RCCL) + fused AdamW step + LR update.  Workload at N=1: BASELINE
config 2 (Segformer-B0 <- B2, CGD group 8, T 4, bs 8, 512x512, 150 classes), fp32 as the reference.

After the warm-up the timed region (EXACTLY K steps, barrier + synchronize on both sides, max over ranks) is run --repeats times (3) back
to back: `value` / `ms_per_step` are the MEDIAN region, `value_min` / `value_max` the slowest / fastest, so a box's spread is in the line.
Rank 0 prints ONE SHORT JSON line (< 3.5 KB: the driver keeps only the tail of stdout) with, besides the contract fields:
  roofline      the HBM-bound CGD kernels (R1 fwd+bwd) at the config-2 operand shape, timed with HIP
                events in this same process: achieved = 5*N*4 bytes / (t_fwd + t_bwd);
  cpu_baseline  the whole KD step on host cores (networks on torch-CPU, criteria from oracle/), B=2,
                a bounded sample (N=1 only);
  config.arithmetic / config.value_exact_f32   the shipped fp32 mode computes its GEMM-shaped products as split-bf16
                (fp32-grade by test); the same workload on exact-f32 MFMA is measured by a child run (N=1 only).
  config.value_deterministic   the same workload under `--deterministic` (the reference's only launch mode, tools/dist_train.sh:8), child run.
--kernel-rooflines additionally writes the per-kernel-family table to a side file (roofline.kernels_file).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)


def build_model(cfg, device):
    import segdistill_amd
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.segmentors import sd_module
    segdistill_amd.register_all()
    sd_module.SYNTHETIC_WEIGHTS_OK = True  # no checkpoints offline: random-init weights of the named architectures
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = build_segmentor(cfg.model.to_dict() if hasattr(cfg.model, 'to_dict') else dict(cfg.model))
    return model.to(device)


def timed_steps(trainer, data, steps, world):
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cur = data.next()
    for _ in range(steps):
        nxt = data.next()            # the data pipeline knows the next batch: the frozen teacher runs one batch ahead
        trainer.step(cur, nxt)
        cur = nxt
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    return time.perf_counter() - t0


def traffic_from_counters(d, algorithmic_bytes=5 * 8 * 150 * 512 * 512 * 4):
    """HBM bytes of one R1 forward + backward from a rocprofv3 PMC dump (tools/pmc_summary.py's JSON, or a published summary that keeps
    it under `raw_counters`).  Corrections exactly as MI355X_MICROARCH.md (HBM section) prescribes: counter unit 1024 B; FETCH_SIZE tallies the
    128-B requests of a 16-B-per-lane streaming read at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
    Returns the summary dict (the keys profiles/traffic_rNN.json carries); raises KeyError when the dump lacks the R1 kernels."""
    if isinstance(d.get('cgd_kl_r1_fwd_bwd_bytes'), (int, float)):
        return d
    raw = d.get('raw_counters', d)

    def one(prefix):
        ks = [k for k in raw if k.startswith(prefix)]
        if len(ks) != 1:
            raise KeyError(prefix)
        c = raw[ks[0]]
        return (2 * c['FETCH_SIZE']['mean'] + c['WRITE_SIZE']['mean']) * 1024.0

    fwd, bwd = one('sd::cgd_fwd_partials<'), one('sd::cgd_bwd<')
    return {'raw_counters': raw, 'cgd_kl_r1_fwd_bytes': fwd, 'cgd_kl_r1_bwd_bytes': bwd, 'cgd_kl_r1_fwd_bwd_bytes': fwd + bwd,
            'algorithmic_bytes': algorithmic_bytes, 'ratio_traffic_over_algorithmic': (fwd + bwd) / algorithmic_bytes}


def source_fingerprint():
    """sha1 over the kernel sources and the Python package: what a stored per-step kernel table (profiles/rNN_step_top5.json) was measured
    on.  There is no .git on the GPU box, so the commit cannot be asked for there; the same bytes give the same fingerprint everywhere."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, 'segdistill_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'segdistill_amd', 'csrc', '*.h'))
                   + glob.glob(os.path.join(ROOT, 'segdistill_amd', '**', '*.py'), recursive=True))
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def stored_step_top5():
    """The newest profiles/r*_step_top5.json IF it was measured on this very source tree (its `fingerprint` equals source_fingerprint());
    a table of an older tree is dropped rather than quoted next to numbers it no longer describes."""
    top = _newest_profile('r*_step_top5.json')
    if not top:
        return None
    d = json.load(open(top))
    if not isinstance(d, dict) or d.get('fingerprint') != source_fingerprint():
        return None
    return {'source': 'stored: ' + os.path.relpath(top, ROOT), 'fingerprint': d['fingerprint'], 'commit': d.get('commit'), 'kernels': d['kernels'][:5]}


def roofline_leg(device, B, C=150, HW=512, g=8, tau=4.0, reps=20):
    """R1 CGD kernels on operands of the config-2 softmax shape, HIP events on torch's current stream
    (the stream the C ABI launches on)."""
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=device).manual_seed(1234)
    S = 2 * torch.randn(B, C, HW, HW, device=device, generator=gen)
    T = 2 * torch.randn(B, C, HW, HW, device=device, generator=gen)
    rows = B * (-(-C // g))
    row_lse2 = torch.empty(rows, 2, device=device)
    row_kl = torch.empty(rows, device=device)
    loss = torch.empty((), device=device)
    dS = torch.empty_like(S)
    wsb = L.sd_cgd_kl_workspace_bytes(B, C, HW, HW, g)
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        rc = L.sd_cgd_kl_fwd(S.data_ptr(), T.data_ptr(), 0, B, C, HW, HW, g, 1 / tau, 3.0 / rows, None, row_lse2.data_ptr(),
                             row_kl.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, st)
        assert rc == 0, rc

    def bwd():
        rc = L.sd_cgd_kl_bwd(S.data_ptr(), T.data_ptr(), 0, B, C, HW, HW, g, 1 / tau, 3.0 / (rows * tau), None, row_lse2.data_ptr(),
                             None, dS.data_ptr(), st)
        assert rc == 0, rc

    def avg_ms(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    tf, tb = avg_ms(fwd), avg_ms(bwd)
    N = S.numel()
    achieved = 5 * N * 4 / ((tf + tb) * 1e-3) / 1e9
    # HBM traffic of one fwd+bwd from the PMC counters.  NOT measured by this run (PMC collection needs rocprofv3): it is read from the
    # newest profiles/traffic_rNN.json, which tools/refresh_profiles.sh produced from separate --pmc FETCH_SIZE / WRITE_SIZE passes over
    # tools/kernel_rooflines.py with THESE kernels; `traffic_source` names the file (null when there is none).
    traffic, traffic_source = None, None
    import glob
    for tpath in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'traffic_r*.json')), reverse=True):
        try:
            traffic = traffic_from_counters(json.load(open(tpath)))['cgd_kl_r1_fwd_bwd_bytes']
            traffic_source = 'stored: ' + os.path.relpath(tpath, ROOT)
            break
        except Exception:
            traffic = None
    del S, T, dS
    return {
        'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
        'traffic': traffic, 'traffic_source': traffic_source,
        'kernel': 'cgd_fwd_partials + cgd_bwd (R1, operands at softmax resolution)',
        'operand_shape': [B, C, HW, HW], 'algorithmic_bytes': 5 * N * 4,
        'fwd_ms': round(tf, 4), 'bwd_ms': round(tb, 4),
        'fwd_GBps': round(2 * N * 4 / (tf * 1e-3) / 1e9, 1), 'bwd_GBps': round(3 * N * 4 / (tb * 1e-3) / 1e9, 1),
    }


def fused_leg(device, B, C=150, hw=128, F=4, g=8, tau=4.0, reps=20):
    """R2 (fused-upsample) kernels at the config-2 tap shape: effective GB/s by the R1 definition."""
    from segdistill_amd import _lib
    L = _lib.lib()
    gen = torch.Generator(device=device).manual_seed(1234)
    s = 2 * torch.randn(B, C, hw, hw, device=device, generator=gen)
    t = 2 * torch.randn(B, C, hw, hw, device=device, generator=gen)
    H = hw * F
    rows = B * (-(-C // g))
    row_lse2 = torch.empty(rows, 2, device=device)
    row_kl = torch.empty(rows, device=device)
    loss = torch.empty((), device=device)
    ds = torch.empty_like(s)
    wsb = L.sd_cgd_kl_up_workspace_bytes(B, C, hw, hw, H, H, g)
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        rc = L.sd_cgd_kl_up_fwd(s.data_ptr(), t.data_ptr(), 0, B, C, hw, hw, H, H, g, 1 / tau, 3.0 / rows, None, row_lse2.data_ptr(),
                                row_kl.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, st)
        assert rc == 0, rc

    def bwd():
        rc = L.sd_cgd_kl_up_bwd(s.data_ptr(), t.data_ptr(), 0, B, C, hw, hw, H, H, g, 1 / tau, 3.0 / (rows * tau), None,
                                row_lse2.data_ptr(), None, ds.data_ptr(), st)
        assert rc == 0, rc

    out = {}
    for name, fn in (('fwd_ms', fwd), ('bwd_ms', bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / reps, 4)
    N = B * C * H * H
    out['effective_GBps_r1_definition'] = round(5 * N * 4 / ((out['fwd_ms'] + out['bwd_ms']) * 1e-3) / 1e9, 1)
    out['tap_bytes'] = 5 * s.numel() * 4
    return out


def kernel_roofline_entries(groups=('r2', 'r1_bf16', 'tok', 'align', 'ce', 'pix', 'at', 'ifvd', 'sra', 'optim', 'dw', 'ln', 'upsum', 'resize', 'gemm', 'pred',
                                    'wgrad_bf16', 'ppm', 'wattn'), reps=10):
    """tools/kernel_rooflines.py, one CHILD process per kernel family.  Called only from a parent that has not touched the GPU yet
    (main() runs it before init_distributed / any torch.cuda call) and only at N = 1.  A family that times out, dies or writes bad
    JSON becomes one error entry; nothing here can take the bench line down."""
    import subprocess
    import tempfile
    out = []
    for g in groups:
        try:
            with tempfile.NamedTemporaryFile(suffix='.json') as f:
                r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'kernel_rooflines.py'), '--only', g, '--reps', str(reps), '--json', f.name],
                                   cwd=ROOT, capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    out.append({'name': g, 'error': f'kernel_rooflines.py --only {g} exited with {r.returncode}'})
                    continue
                out += json.load(open(f.name))
        except (subprocess.TimeoutExpired, OSError, ValueError) as e:
            out.append({'name': g, 'error': f'{type(e).__name__}: {str(e)[:120]}'})
    return out


def variant_child(argv_base, steps, warmup, env_extra=None, flags=(), timeout=420, repeats=1):
    """The same workload in a CHILD process started before this one touches the GPU, with a switch flipped: every split-bf16 product back on
    exact-f32 MFMA (SEGDISTILL_SPLIT_BF16=0 -> config.value_exact_f32), or the reference's launch mode `--deterministic`
    (tools/dist_train.sh:8 -> config.value_deterministic).  Returns (imgs/s or None, note)."""
    import subprocess
    env = dict(os.environ, SEGDISTILL_BENCH_CHILD='1', **(env_extra or {}))
    cmd = [sys.executable, os.path.abspath(__file__)] + argv_base + ['--steps', str(steps), '--warmup', str(warmup), '--no-roofline',
                                                                     '--no-cpu-baseline', '--no-exact-f32', '--no-deterministic-child',
                                                                     '--repeats', str(repeats)] + list(flags)
    try:
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
        if r.returncode != 0 or not lines:
            return None, f'child exited with {r.returncode}'
        return float(json.loads(lines[-1])['value']), None
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
        return None, f'{type(e).__name__}: {str(e)[:120]}'


def exact_f32_child(argv_base, steps, warmup, timeout=420):
    return variant_child(argv_base, steps, warmup, env_extra={'SEGDISTILL_SPLIT_BF16': '0'}, timeout=timeout)


def deterministic_child(argv_base, steps, warmup, timeout=420):
    # the parent's own step count, warm-up and median-of-three: the figure is meant to be read next to `value` (a 10-step single region after a
    # 4-step warm-up read 3 % low on a box where the two modes are level)
    return variant_child(argv_base, steps, warmup, flags=('--deterministic',), timeout=timeout, repeats=3)


def _usable_cores():
    """Host cores this process may really use: min(affinity mask, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline_leg(cfg, batch=2, budget_s=25.0, max_threads=32):
    """Whole KD step on host cores: networks on torch-CPU, criteria = oracle eager restatement.
    Bounded: one warm-up step, then timed steps until ~budget_s of CPU work is spent (at least 1)."""
    from oracle.eager_modules import swap_in_eager_criteria
    from segdistill_amd.engine import KDTrainer, SyntheticADE
    threads = max(1, min(_usable_cores(), max_threads))  # torch-CPU stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = build_model(cfg, torch.device('cpu'))
    swap_in_eager_criteria(model)
    trainer = KDTrainer(model, cfg.optimizer.to_dict() if hasattr(cfg.optimizer, 'to_dict') else dict(cfg.optimizer),
                        dict(cfg.lr_config), world=1)
    data = SyntheticADE(batch, device='cpu', pool=1)
    t0 = time.perf_counter()
    trainer.step(data.next())  # warm-up
    warm = time.perf_counter() - t0
    timed = max(1, min(5, int(budget_s / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(timed):
        trainer.step(data.next())
    dt = time.perf_counter() - t0
    return {'value': round(batch * timed / dt, 4), 'unit': 'imgs/s', 'cores': threads, 'kind': 'port',
            'sample': f'{timed} KD step(s) of the same config at batch {batch} after 1 warm-up ({warm:.1f} s): networks on torch-CPU fp32, '
                      f'criteria from oracle/ (losses.py:95-113 restated)',
            's_per_step': round(dt / timed, 3)}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no torchrun environment: THIS process becomes the launcher (the reference's
    tools/dist_train.sh:8 is `python -m torch.distributed.launch --nproc_per_node=$GPUS ... train.py`).  It has not touched the GPU
    (importing torch does not initialise HIP; nothing below does either), starts `torch.distributed.run` with one rank per GPU as a
    CHILD process -- never an exec --, relays its output (rank 0's JSON line) and exits with its status."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['SEGDISTILL_BENCH_SPAWNED'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print(f'[bench] launching {n} ranks: {" ".join(cmd[1:9])} bench.py ...', file=sys.stderr)
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def _newest_profile(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)), reverse=True)
    return files[0] if files else None


def compose_line(*, args, world, B, dt, rank_ms, graphed, segments, trainer_bf16, arithmetic, grad_bytes, ranks_seen, backend, rccl_version,
                 logs, allreduce_ms=None, syncbn_ms=None, roofline=None, cpu_baseline=None, exact_f32=None, kernels_file=None, errors=None,
                 dts=None, deterministic=None, value_deterministic=None, allreduce_exposed_ms=None, peak_mem_gb=None):
    """The ONE JSON line of the contract, kept SHORT (the driver keeps only the tail of stdout: round 2's 24 KB line was lost).  Pure: the
    CPU test tests/test_bench_line_cpu.py builds it from canned leg outputs and holds it under 3500 bytes."""
    cfg_name = os.path.basename(args.config)
    is_cfg2 = 'cfg2' in cfg_name
    # `dt` is the MEDIAN of the timed regions (each exactly `steps` steps, barrier + synchronize on both sides, max over ranks); `dts` lists
    # them all, so the spread of the box rides along in the one record that counts
    dts = sorted(dts) if dts else [dt]
    line = {
        'metric': 'imgs/sec/node KD train_step, ' + ('Segformer-B2->B0 512x512' if is_cfg2 else cfg_name),
        'value': round(world * B * args.steps / dt, 3), 'unit': 'imgs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16' if trainer_bf16 else 'f32', 'data': 'synthetic',
        'config': {'workload': ('BASELINE configs[1]: Segformer-B0 student + B2 teacher, CGD group=8 T=4 alpha=3, 512x512, 150 classes'
                                if is_cfg2 else cfg_name),
                   'config_file': os.path.relpath(args.config, ROOT), 'per_gpu_batch': B, 'global_batch': B * world, 'parallelism': f'dp{world}',
                   'arithmetic': arithmetic, 'kd_path': args.kd_path, 'hip_graph': graphed, 'graph_segments': segments,
                   'weights': 'random-init', 'grad_allreduce_bytes': grad_bytes, 'rccl_ranks': ranks_seen, 'dist_backend': backend,
                   'rccl_version': rccl_version, 'rank_ms_per_step': rank_ms},
        'final_log_vars': {k: round(v, 5) for k, v in list(logs.items())[:8]},
    }
    if len(dts) > 1:
        line['value_min'] = round(world * B * args.steps / dts[-1], 3)
        line['value_max'] = round(world * B * args.steps / dts[0], 3)
        line['timed_regions'] = len(dts)
    if deterministic:
        line['config']['deterministic'] = True
    if peak_mem_gb is not None:
        line['config']['peak_mem_gb'] = peak_mem_gb       # torch.cuda.max_memory_allocated over warm-up + timed regions (this rank)
    if value_deterministic is not None:
        line['config']['value_deterministic'] = value_deterministic
    if allreduce_ms is not None:
        line['config']['grad_allreduce_ms'] = allreduce_ms
    if allreduce_exposed_ms is not None:
        line['config']['grad_allreduce_exposed_ms'] = allreduce_exposed_ms
    if syncbn_ms is not None:
        line['config']['syncbn_collectives'] = syncbn_ms          # device time of the norms' statistics exchanges between the step's graphs
    if exact_f32 is not None:
        line['config']['value_exact_f32'] = exact_f32
    if roofline is not None:
        line['roofline'] = roofline
        if kernels_file:
            line['roofline']['kernels_file'] = kernels_file
    if cpu_baseline is not None:
        line['cpu_baseline'] = cpu_baseline
    if errors:
        line['errors'] = [e[:160] for e in errors][:4]
    return line


def allreduce_probe(trainer, reps=10):
    """Device time of the step's gradient exchange (pack + ONE all-reduce of the flat buffer), HIP events on the stream it runs on.
    COLLECTIVE: every rank calls it."""
    r = trainer.reducer
    if not r.collective:
        return None
    for _ in range(2):
        r.pack()
        r.exchange()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r.pack()
        r.exchange()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


def ranks_agree(ok):
    """True only if EVERY rank says ok (one tiny all-reduce; no process group: this rank's word)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(ok)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def choose_graph_mode(trainer, next_batch, requested, errors):
    """'full' -> 'hybrid' -> False (eager), the first mode that EVERY rank could set up (requested: auto | on | hybrid | off; `on` and
    `hybrid` try that mode only).  A rank whose capture succeeded while another rank's failed drops its graphs again: all ranks of a
    data-parallel job step the same way, and the line says which mode that was and why the better ones were not taken."""
    plan = {'auto': ['full', 'hybrid'], 'on': ['full'], 'hybrid': ['hybrid'], 'off': []}[requested]
    for mode in plan:
        try:
            ok = trainer.enable_graph(next_batch()) if mode == 'full' else trainer.enable_hybrid_graph(next_batch())
        except Exception as e:  # noqa: BLE001 -- enable_* catch their own failures; belt and braces: the line must survive
            ok = False
            trainer.graph_error = f'{type(e).__name__}: {e}'
        if ranks_agree(ok):
            return mode
        why = getattr(trainer, 'graph_error', None) if not ok else 'another rank could not capture'
        errors.append(f'hip_graph {mode}: {why}')
        trainer.disable_graph()
    return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default=os.path.join(ROOT, 'configs', 'kd', 'cfg2_segformer_b2_b0_cgd.py'))
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: config data.samples_per_gpu)')
    ap.add_argument('--kd-path', choices=['fused', 'r1'], default='fused',
                    help="fused: bilinear resize fused into the CGD kernels (R2); r1: ATen resize + streaming kernels")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-exact-f32', action='store_true', help='skip the exact-f32 A/B child run (config.value_exact_f32)')
    ap.add_argument('--exact-f32-steps', type=int, default=0, help='timed steps of the exact-f32 A/B child (default: min(steps, 10))')
    ap.add_argument('--repeats', type=int, default=3,
                    help='timed regions of EXACTLY --steps steps each, back to back; value / ms_per_step = the median region, value_min / value_max '
                         'the slowest / fastest one')
    ap.add_argument('--deterministic', action='store_true',
                    help="the reference's launch mode (tools/dist_train.sh:8): engine.set_deterministic -- MIOpen held to its deterministic algorithms")
    ap.add_argument('--no-deterministic-child', action='store_true', help='skip the --deterministic child run (config.value_deterministic)')
    ap.add_argument('--kernel-rooflines', action='store_true',
                    help='N=1 only: also time every hand-written kernel family against its own bound (tools/kernel_rooflines.py, child processes '
                         'started before this process touches the GPU); the table goes to a SIDE FILE named by roofline.kernels_file')
    ap.add_argument('--kernels-out', default=os.path.join(ROOT, 'gpurun_out', 'bench_kernels.json'))
    ap.add_argument('--cpu-threads', type=int, default=32)
    ap.add_argument('--graph', choices=['auto', 'on', 'hybrid', 'off'], default='auto',
                    help='on: capture the whole forward+backward as hipGraphs (cut at the SyncBN collectives when ranks > 1); hybrid: '
                         'capture only the teacher forward and the student backbone fwd/bwd; auto: on, falling back to hybrid')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    # ---- child processes first: nothing above or below this block has touched the GPU in THIS process yet ------------------------------
    errors = []
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    single = world_env == 1 and args.gpus == 1 and os.environ.get('SEGDISTILL_BENCH_CHILD') != '1'
    kernels_file = None
    if single and args.kernel_rooflines and not args.no_roofline:
        try:
            entries = kernel_roofline_entries()
            os.makedirs(os.path.dirname(args.kernels_out), exist_ok=True)
            with open(args.kernels_out, 'w') as f:
                json.dump(entries, f, indent=0)
            kernels_file = os.path.relpath(args.kernels_out, ROOT)
        except Exception as e:  # noqa: BLE001
            errors.append(f'kernel rooflines: {type(e).__name__}: {e}')
    exact_f32 = None
    split_on = os.environ.get('SEGDISTILL_SPLIT_BF16', '1') == '1'
    if single and split_on and not args.no_exact_f32:
        base = ['--config', args.config, '--kd-path', args.kd_path, '--graph', args.graph] + (['--batch', str(args.batch)] if args.batch else [])
        # a SHORT child (a second full benchmark doubled the wall time under the driver's budget and heated the GPU in front of
        # the headline run); tools/refresh_profiles.sh asks for the full-length A/B with --exact-f32-steps
        exact_f32, err = exact_f32_child(base, args.exact_f32_steps or min(args.steps, 10), min(args.warmup, 4))
        if err:
            errors.append('exact-f32 A/B: ' + err)
    value_deterministic = None
    if single and not args.deterministic and not args.no_deterministic_child:
        base = ['--config', args.config, '--kd-path', args.kd_path, '--graph', args.graph] + (['--batch', str(args.batch)] if args.batch else [])
        value_deterministic, err = deterministic_child(base, args.steps, args.warmup)
        if err:
            errors.append('deterministic child: ' + err)

    from segdistill_amd.config import Config
    from segdistill_amd.engine import KDTrainer, SyntheticADE, init_distributed, set_deterministic
    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit(f'[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    torch.backends.cudnn.benchmark = True
    if args.deterministic:
        set_deterministic(True)

    cfg = Config.fromfile(args.config)
    B = args.batch or int(cfg.data.samples_per_gpu)
    torch.manual_seed(0)
    model = build_model(cfg, device)
    if args.kd_path == 'r1':
        for c in model.distillation_loss.criteria:
            if hasattr(c, 'fuse_resize'):
                c.fuse_resize = False
    opt_cfg = cfg.optimizer.to_dict() if hasattr(cfg.optimizer, 'to_dict') else dict(cfg.optimizer)
    trainer = KDTrainer(model, opt_cfg, dict(cfg.lr_config), max_iters=int(cfg.runner.max_iters), world=world,
                        precision=cfg.get('precision'))
    data = SyntheticADE(B, size=tuple(cfg.get('crop_size', (512, 512))), num_classes=int(cfg.get('num_classes', 150)), seed=0,
                        rank=rank, device=device)
    graphed = False
    n_eager_warm = max(1, args.warmup // 2)
    for _ in range(n_eager_warm):            # eager warm-up first (MIOpen find, hipBLASLt heuristics, allocator)
        trainer.step(data.next())
    # auto: the whole step as hipGraphs at any world size (with ranks > 1 the capture is cut at the SyncBN collectives,
    # engine/segments.py); a capture that fails ON ANY RANK degrades every rank to the hybrid mode, then to eager
    graphed = choose_graph_mode(trainer, data.next, args.graph, errors)
    replay_failed = None
    try:
        for _ in range(args.warmup - n_eager_warm):
            trainer.step(data.next())
    except Exception as e:  # noqa: BLE001 -- a replay that fails (first RCCL run inside a segmented capture) must not cost the line
        replay_failed = f'{type(e).__name__}: {e}'
    if world > 1 and replay_failed is not None:
        # a rank that raised in the middle of a step has not issued the step's remaining collectives: its peers are blocked in them, and a
        # rank that carried on alone (eager) would never meet them again -- say why and leave; never re-exec, never hang
        print(f'[bench] rank {rank}: {graphed} graph replay failed in warm-up ({replay_failed}); exiting', file=sys.stderr)
        sys.stderr.flush()
        os._exit(3)
    if replay_failed is not None:
        errors.append(f'{graphed} graph replay failed in warm-up ({replay_failed}); eager steps instead')
        trainer.disable_graph()
        graphed = False
        for _ in range(max(1, args.warmup - n_eager_warm)):
            trainer.step(data.next())
    # the timed regions: each EXACTLY args.steps steps between barrier + synchronize, max over ranks; the line's value is the median region
    dts, rank_ms = [], None
    for _ in range(max(1, args.repeats)):
        dt_local = timed_steps(trainer, data, args.steps, world)
        t = torch.tensor([dt_local], device=device, dtype=torch.float64)
        if dist.is_initialized():
            every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
            dist.all_gather(every, t)
            per = [float(x.item()) / args.steps * 1e3 for x in every]
            rank_ms = {'min': round(min(per), 3), 'max': round(max(per), 3)}
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dts.append(float(t.item()))
    dt = sorted(dts)[len(dts) // 2]
    peak_mem_gb = round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 3)
    logs = trainer.log_values()
    # how many ranks really took part: an all-reduce of ones over the process group the gradients went through
    ranks_seen, backend = 1, None
    if dist.is_initialized():
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        ranks_seen, backend = int(ones.item()), dist.get_backend()
        assert ranks_seen == dist.get_world_size() == world
    try:
        rccl_version = '.'.join(str(v) for v in torch.cuda.nccl.version()) if backend == 'nccl' else None
    except Exception:  # noqa: BLE001
        rccl_version = None
    allreduce_ms = allreduce_probe(trainer) if dist.is_initialized() else None       # collective: all ranks
    syncbn_ms = None
    if dist.is_initialized() and graphed == 'full' and getattr(trainer, '_seg', None) is not None and trainer._seg.cuts:
        # the statistics exchanges of the synchronised norms between the step's graphs: 3 more steps with HIP events around each (every rank)
        trainer._seg.time_collectives = True
        for _ in range(3):
            trainer.step(data.next())
        ms, n = trainer._seg.collective_ms()
        trainer._seg.time_collectives = False
        syncbn_ms = {'ms_per_step': round(ms, 4), 'collectives_per_step': n}
    if rank == 0 and allreduce_ms is not None:
        print(f'[bench] gradient exchange (pack + one all-reduce of {trainer.reducer.nbytes} bytes over {ranks_seen} ranks, {backend}): '
              f'{allreduce_ms} ms per step by HIP events; step {dt / args.steps * 1e3:.3f} ms; hip_graph={graphed}', file=sys.stderr)

    if rank == 0:
        segs = (len([g for g in trainer._seg.items if isinstance(g, torch.cuda.CUDAGraph)]) if graphed == 'full' else None)
        arithmetic = ('bf16 storage, fp32 accumulate' if trainer.bf16 else
                      ('split-bf16 (bf16x3 on the bf16 MFMA pipe, fp32-grade: tests/test_token_gemm_gpu.py)' if split_on else 'exact-f32 MFMA'))
        roofline = cpu = None
        try:
            if not args.no_roofline:
                torch.cuda.empty_cache()
                roofline = roofline_leg(device, B)
                roofline['fused_r2'] = fused_leg(device, B)
                top = stored_step_top5()
                if top:
                    roofline['step_top5'] = top
        except Exception as e:  # noqa: BLE001 -- an optional leg never costs the headline
            errors.append(f'roofline leg: {type(e).__name__}: {e}')
        try:
            if world == 1 and not args.no_cpu_baseline:
                cpu = cpu_baseline_leg(cfg, max_threads=args.cpu_threads)
        except Exception as e:  # noqa: BLE001
            errors.append(f'cpu baseline leg: {type(e).__name__}: {e}')
        line = compose_line(args=args, world=world, B=B, dt=dt, rank_ms=rank_ms, graphed=graphed, segments=segs, trainer_bf16=trainer.bf16,
                            arithmetic=arithmetic, grad_bytes=trainer.reducer.nbytes, ranks_seen=ranks_seen, backend=backend,
                            rccl_version=rccl_version, logs=logs, allreduce_ms=allreduce_ms, syncbn_ms=syncbn_ms, roofline=roofline, cpu_baseline=cpu,
                            exact_f32=exact_f32, kernels_file=kernels_file, errors=errors, dts=dts, deterministic=args.deterministic,
                            value_deterministic=value_deterministic, peak_mem_gb=peak_mem_gb)
        sys.stderr.flush()
        print(json.dumps(line))
        sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

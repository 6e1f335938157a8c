"""RCCL on the GPU box with ONE rank (SEGDISTILL_FORCE_COLLECTIVES=1): the process group initialises over the `nccl`
backend, the flat gradient all-reduce runs through RCCL every step, and both hipGraph modes capture and replay next to
RCCL's watchdog thread.  (Two ranks cannot share one GPU under RCCL -- tests/test_two_ranks_gpu.py covers world 2 via
gloo; this file covers the RCCL side of the same code path.)"""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _bench(graph, forced):
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('SEGDISTILL_FORCE_COLLECTIVES', None)
    if forced:
        env['SEGDISTILL_FORCE_COLLECTIVES'] = '1'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--batch', '2', '--steps', '3', '--warmup', '2', '--graph', graph,
           '--no-cpu-baseline', '--no-roofline']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize('graph', ['hybrid', 'on', 'off'])
def test_rccl_all_reduce_and_graph_capture_coexist(graph):
    plain, _ = _bench(graph, forced=False)
    forced, err = _bench(graph, forced=True)
    assert 'capture failed' not in err, err[-3000:]
    want = {'hybrid': 'hybrid', 'on': 'full', 'off': False}[graph]
    assert forced['config']['hip_graph'] == want and plain['config']['hip_graph'] == want
    # same seeds, same data, all-reduce over one rank is the identity: the logged losses agree (atomics-free kernels; the
    # library GEMMs may pick different algorithms run to run, hence a tolerance)
    for k, v in plain['final_log_vars'].items():
        assert math.isfinite(forced['final_log_vars'][k])
        tol = 1.0 if 'acc' in k else 2e-3 * max(1.0, abs(v))
        assert forced['final_log_vars'][k] == pytest.approx(v, abs=tol), (k, v, forced['final_log_vars'][k])

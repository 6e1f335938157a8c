"""The multi-rank GPU path end to end on a ONE-GPU box: two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on
one device), which exercises everything a real 2-GPU run does except the RCCL transport: SyncBatchNorm through the process
group, the flat gradient all-reduce, the packed log all-reduce, rank-0 broadcast, the hybrid hipGraph mode and bench.py's
max-over-ranks timing / JSON contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize('config,batch,mode,segments,ranks', [
    ('cfg2_segformer_b2_b0_cgd.py', 1, 'full', 3, 2),     # one SyncBN in the student: whole step replayed, cut at its two collectives
    # four ranks on the one GPU (the box admits at most 6 processes on its card, so 8 cannot be rehearsed here): ports, segment count,
    # the per-rank spread and the collective sizes must not depend on the rank count
    ('cfg2_segformer_b2_b0_cgd.py', 1, 'full', 3, 4),
])
def test_bench_ranks_on_one_gpu(config, batch, mode, segments, ranks):
    env = dict(os.environ, SEGDISTILL_DIST_BACKEND='gloo', SEGDISTILL_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import bench
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(bench._free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--steps', '3', '--warmup', '4', '--batch', str(batch),
           '--no-roofline', '--config', os.path.join(ROOT, 'configs', 'kd', config)]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d['n_gpus'] == ranks and d['config']['global_batch'] == ranks * batch and d['config']['parallelism'] == f'dp{ranks}'
    assert d['config']['rccl_ranks'] == ranks and d['config']['grad_allreduce_bytes'] == 15010776      # the B0 student's trainable fp32 parameters
    assert d['config']['hip_graph'] == mode and d['config']['graph_segments'] == segments and d['scaling'] == 'weak' and d['value'] > 0
    assert all(v == v and abs(v) < 1e6 for v in d['final_log_vars'].values())
    assert 'cpu_baseline' not in d                       # N=1 only
    assert len(lines[0]) < 2500 and d['config']['rank_ms_per_step']['min'] <= d['config']['rank_ms_per_step']['max']
    assert d['config']['grad_allreduce_ms'] > 0 and 'gradient exchange' in res.stderr


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment (how the driver may invoke it): the process must launch the two ranks
    itself -- before touching the GPU -- and relay rank 0's line; reporting n_gpus = 1 here was round 1's defect."""
    env = dict(os.environ, SEGDISTILL_DIST_BACKEND='gloo', SEGDISTILL_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '2', '--batch', '1', '--no-roofline',
           '--graph', 'off']
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['rccl_ranks'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['dist_backend'] == 'gloo'

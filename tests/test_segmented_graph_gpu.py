"""The whole-step hipGraph capture with MORE THAN ONE RANK (engine/segments.py): the capture is cut at the student's
SyncBatchNorm collectives and replayed as graph | all_gather | graph | all_reduce | graph.

* two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device): the segmented step must train like
  the eager step that uses torch's own SyncBatchNorm, on both ranks, and keep the replicas identical;
* one rank over RCCL with the synchronised layer forced on: the same chain with the real RCCL collectives issued
  between graph replays."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', '_segmented_worker.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _results(res):
    """The workers' `RESULT {json}` records; two ranks share one pipe, so a record may be followed by the other rank's on the same line."""
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out, dec, pos = [], json.JSONDecoder(), 0
    while True:
        pos = res.stdout.find('RESULT ', pos)
        if pos < 0:
            return out
        obj, end = dec.raw_decode(res.stdout, pos + len('RESULT '))
        out.append(obj)
        pos = end


def _check(r):
    assert r['chained'] and r['graph'], r.get('warnings')
    assert r['graphs'] == 3 and r['cuts'] == 2          # forward | statistics | rest of forward + backward | dy sums | backward
    assert r['cnt_after_capture'] == 0 and r['tracked_after_capture'] == 0     # neither the step counter nor the BN buffers moved
    assert r['running_stats_moved_by_capture'] == 0.0
    for it, (ve, vg) in enumerate(r['steps']):
        assert list(ve) == list(vg)
        for k in ve:
            tol = 100.0 * 8 / (2 * 128 * 128) if 'acc' in k else 3e-4 * max(1.0, abs(ve[k]))
            assert vg[k] == pytest.approx(ve[k], abs=tol), (it, k, ve[k], vg[k])
    assert r['param_rel_l2'] < 3e-4
    assert r['running_mean_diff'] < 1e-4 and r['running_var_rel'] < 1e-3
    assert r['tracked'][0] == r['tracked'][1]


def test_segmented_graph_two_ranks_gloo():
    env = dict(os.environ, SEGDISTILL_DIST_BACKEND='gloo', SEGDISTILL_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), WORKER]
    out = _results(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert sorted(r['rank'] for r in out) == [0, 1]
    for r in out:
        assert r['world'] == 2
        _check(r)
    assert out[0]['digest'] == out[1]['digest']          # replicas stay identical
    assert out[0]['steps'] == out[1]['steps']            # log values are rank means


def test_segmented_graph_with_kd_tap_upstream_of_the_norm():
    """Config 5's situation: a KD tap on decode_head.linear_c1 puts backbone / linear_c1 parameters into BOTH autograd walks of the
    segmented backward.  Round 1 ran both walks inside one deferred scope and silently dropped the second walk's contribution for
    every deferred parameter gradient (LayerNorm / Linear / depth-wise weights); the segmented step must match the eager step."""
    env = dict(os.environ, SEGDISTILL_DIST_BACKEND='gloo', SEGDISTILL_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), WORKER, 'upstream']
    out = _results(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert sorted(r['rank'] for r in out) == [0, 1]
    for r in out:
        assert len([k for k in r['steps'][0][0] if k.startswith('loss_')]) == 2
        assert any(k.startswith('loss_decode_head.linear_c1') and v[0][k] > 1e-3 for v in r['steps'] for k in v[0])   # a live KD term
        _check(r)
    assert out[0]['digest'] == out[1]['digest']


def test_segmented_graph_one_rank_rccl():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0', SEGDISTILL_FORCE_COLLECTIVES='1', SEGDISTILL_FORCE_SYNCBN='1')
    env.pop('SEGDISTILL_DIST_BACKEND', None)
    out = _results(subprocess.run([sys.executable, WORKER], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert len(out) == 1 and out[0]['world'] == 1
    _check(out[0])


def test_resnet_student_two_ranks_falls_back_cleanly():
    """28 SyncBN layers in the student: the segmented capture declines (before any capture starts), the hybrid mode graphs the teacher
    only -- no collective is ever recorded -- and both ranks keep training in step."""
    import math
    env = dict(os.environ, SEGDISTILL_DIST_BACKEND='gloo', SEGDISTILL_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), WORKER, 'pspnet']
    out = _results(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert sorted(r['rank'] for r in out) == [0, 1]
    for r in out:
        assert r['full'] is False and any('synchronised norms' in w for w in r['warnings'])
        assert r['hybrid'] is True and r['teacher_graphed'] and not r['backbone_graphed']
        assert all(math.isfinite(v) for step in r['steps'] for v in step.values())
    assert out[0]['digest'] == out[1]['digest'] and out[0]['steps'] == out[1]['steps']


def test_resnet_student_two_ranks_whole_step_replay_when_the_limit_is_lifted():
    """SEGDISTILL_MAX_CHAINED_NORMS=64: the same PSPNet-R18 student (28 + 5 synchronised norms) is captured WHOLE -- ~70 graph segments with
    the SyncBN collectives between them -- and both ranks keep training in step with finite losses."""
    import math
    env = dict(os.environ, SEGDISTILL_DIST_BACKEND='gloo', SEGDISTILL_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0', SEGDISTILL_MAX_CHAINED_NORMS='64')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), WORKER, 'pspnet']
    out = _results(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900))
    assert sorted(r['rank'] for r in out) == [0, 1]
    for r in out:
        assert r['full'] is True, r['warnings']
        assert r['segments'] >= 2 * 28 + 1
        assert all(math.isfinite(v) for step in r['steps'] for v in step.values())
    assert out[0]['digest'] == out[1]['digest'] and out[0]['steps'] == out[1]['steps']

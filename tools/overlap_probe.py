"""How much of the look-ahead teacher forward really hides behind the student step?  Without a profiler: wall time per step of (a) the full
graph-replayed step, (b) the student graph + optimizer alone (teacher taps left as they are), (c) the teacher graph alone.
(a) close to (b) + (c) means the two queues hardly overlap: every microsecond saved on EITHER network shortens the step.
usage: python tools/overlap_probe.py [--config configs/kd/cfg2_segformer_b2_b0_cgd.py] [--steps 30]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from segdistill_amd.config import Config  # noqa: E402
from segdistill_amd.engine import KDTrainer, SyntheticADE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--config', default='configs/kd/cfg2_segformer_b2_b0_cgd.py')
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(bench.ROOT, a.config))
torch.manual_seed(0)
model = bench.build_model(cfg, dev)
tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=1, precision=cfg.get('precision'))
data = SyntheticADE(int(cfg.data.samples_per_gpu), device=dev)
for _ in range(3):
    tr.step(data.next())
assert tr.enable_graph(data.next()), tr.graph_error
cur = data.next()
for _ in range(5):
    nxt = data.next()
    tr.step(cur, nxt)
    cur = nxt
torch.cuda.synchronize()
K = a.steps


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


def full():
    global cur
    nxt = data.next()
    tr.step(cur, nxt)
    cur = nxt


def student_only():
    tr._graph.replay()
    tr.optimizer.step()


def teacher_only():
    with torch.cuda.stream(model._side_stream):
        tr._t_graph.replay()


t_full = wall(full)
torch.cuda.synchronize()
t_s = wall(student_only)
model._side_stream.synchronize()
t_t = wall(teacher_only)
model._side_stream.synchronize()
print(f'{os.path.basename(a.config)}: full step {t_full:.3f} ms | student graph + optimizer alone {t_s:.3f} ms | teacher graph alone {t_t:.3f} ms | '
      f'sum {t_s + t_t:.3f} ms -> hidden by overlap {t_s + t_t - t_full:.3f} ms ({100 * (t_s + t_t - t_full) / t_t:.0f} % of the teacher)')

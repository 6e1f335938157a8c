"""Where does the wall time of a graph-replayed KD step go?  Per step: host time inside trainer.step(), GPU time between an
event recorded before the step's first command and one after its last, and the GPU idle between consecutive steps
(start[i+1] - end[i]: positive only when the host is late), without a profiler attached.
usage: python tools/step_gap_probe.py [--graph on|hybrid|off] [--steps 30]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from segdistill_amd.config import Config  # noqa: E402
from segdistill_amd.engine import KDTrainer, SyntheticADE, init_distributed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--graph', default='on')
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--config', default='configs/kd/cfg2_segformer_b2_b0_cgd.py')
a = ap.parse_args()
rank, local, world = init_distributed()
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(bench.ROOT, a.config))
torch.manual_seed(0)
model = bench.build_model(cfg, dev)
tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=world, precision=cfg.get('precision'))
data = SyntheticADE(int(cfg.data.samples_per_gpu), device=dev)
for _ in range(3):
    tr.step(data.next())
if a.graph == 'on':
    assert tr.enable_graph(data.next())
elif a.graph == 'hybrid':
    assert tr.enable_hybrid_graph(data.next())
cur = data.next()
for _ in range(5):
    nxt = data.next()
    tr.step(cur, nxt)
    cur = nxt
torch.cuda.synchronize()
K = a.steps
s = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
e = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
host = []
t00 = time.perf_counter()
for i in range(K):
    nxt = data.next()
    s[i].record()
    t0 = time.perf_counter()
    tr.step(cur, nxt)
    host.append(time.perf_counter() - t0)
    e[i].record()
    cur = nxt
t_enq = time.perf_counter() - t00
torch.cuda.synchronize()
wall = time.perf_counter() - t00
gpu = [s[i].elapsed_time(e[i]) for i in range(K)]
gap = [e[i].elapsed_time(s[i + 1]) for i in range(K - 1)]
med = lambda v: sorted(v)[len(v) // 2]
print(f'graph={a.graph} world={world} collective={tr.reducer.collective}: wall {wall / K * 1e3:.2f} ms/step; host enqueue {t_enq / K * 1e3:.2f} ms/step '
      f'(median step() call {med(host) * 1e3:.2f} ms, max {max(host) * 1e3:.2f}); GPU start->end median {med(gpu):.2f} ms; '
      f'idle between steps median {med(gap) * 1e3:.0f} us, mean {sum(gap) / len(gap) * 1e3:.0f} us')

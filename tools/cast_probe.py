"""Which dtype casts does one KD step of a config issue (eager, after two warm-up steps)?  Prints aten::_to_copy / aten::copy_ calls grouped by
input shape and Python call site -- per-step casts of FROZEN weights are waste (they should come from layers.frozen_derived).
    python tools/cast_probe.py [--config configs/kd/cfg5_segformer_b4_b1_multistage_bf16.py]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from segdistill_amd.config import Config  # noqa: E402
from segdistill_amd.engine import KDTrainer, SyntheticADE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='configs/kd/cfg5_segformer_b4_b1_multistage_bf16.py')
ap.add_argument('--ops', default='aten::_to_copy', help='comma-separated ATen op names to list (e.g. aten::copy_,aten::add,aten::clone,aten::sum)')
ap.add_argument('--by-bytes', action='store_true', help='sort by count x elements instead of count')
ap.add_argument('--all', action='store_true', help='every aten:: op (ignores --ops)')
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(bench.ROOT, a.config))
torch.manual_seed(0)
model = bench.build_model(cfg, dev)
tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=1, precision=cfg.get('precision'))
data = SyntheticADE(int(cfg.data.samples_per_gpu), device=dev)
for _ in range(2):
    tr.step(data.next())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.step(data.next())
    torch.cuda.synchronize()
groups = collections.Counter()
for ev in prof.events():
    if (a.all and ev.name.startswith('aten::') and ev.input_shapes and ev.input_shapes[0]) or ev.name in a.ops.split(','):
        shp = tuple(ev.input_shapes[0]) if ev.input_shapes else ()
        site = next((s for s in (ev.stack or []) if 'segdistill_amd' in s), (ev.stack or ['?'])[0] if ev.stack else '?')
        groups[(ev.name, shp, site.split('/')[-1][:90])] += 1
def _numel(shp):
    n = 1
    for v in shp:
        n *= v
    return n


order = (lambda kv: -kv[1] * _numel(kv[0][1])) if '--by-bytes' in sys.argv else (lambda kv: -kv[1])
for (name, shp, site), n in sorted(groups.items(), key=order)[:60]:
    print(f'{n:4d} x {name:16s} {str(shp):28s} {site}')
print('total:', sum(groups.values()))

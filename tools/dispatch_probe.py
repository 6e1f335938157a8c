"""Which ATen ops that LAUNCH elementwise / copy kernels does one eager KD step issue, and from which line of this package?  A TorchDispatchMode
logs every op in --ops (default: dtype-changing copies) with the innermost segdistill_amd frame of the Python stack (ops issued by the autograd
engine outside a Python backward show as '<engine>').
    python tools/dispatch_probe.py [--config configs/kd/cfg5_segformer_b4_b1_multistage_bf16.py] [--ops copy_,_to_copy,add,add_,mul,clone,cat,sum]"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402
from segdistill_amd.config import Config  # noqa: E402
from segdistill_amd.engine import KDTrainer, SyntheticADE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='configs/kd/cfg5_segformer_b4_b1_multistage_bf16.py')
ap.add_argument('--ops', default='copy_,_to_copy')
ap.add_argument('--min-numel', type=int, default=1024)
ap.add_argument('--graph', action='store_true', help='log what gets CAPTURED into the step hipGraph (trainer.enable_graph) instead of an eager step')
a = ap.parse_args()
want = set(a.ops.split(','))
groups = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in want:
            ts = [t for t in args if isinstance(t, torch.Tensor)]
            o = out if isinstance(out, torch.Tensor) else (ts[0] if ts else None)
            if o is not None and o.is_cuda and o.numel() >= a.min_numel and (not a.graph or torch.cuda.is_current_stream_capturing()):
                if name in ('copy_', '_to_copy') and len(ts) >= 1:
                    src = ts[-1]
                    if name == 'copy_' and len(ts) >= 2 and ts[0].dtype == ts[1].dtype and ts[1].is_contiguous() and ts[0].is_contiguous():
                        tag = 'same-dtype contiguous'
                    else:
                        tag = f'{str(src.dtype)[6:]}->{str(o.dtype)[6:]}' + ('' if src.is_contiguous() else ' strided-src')
                else:
                    tag = str(o.dtype)[6:]
                site = '<engine>'
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if 'segdistill_amd' in fr.filename and 'dispatch_probe' not in fr.filename:
                        site = f'{os.path.relpath(fr.filename, bench.ROOT)}:{fr.lineno}'
                        break
                groups[(name, tag, tuple(o.shape), site)] += 1
        return out


dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(bench.ROOT, a.config))
torch.manual_seed(0)
model = bench.build_model(cfg, dev)
tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=1, precision=cfg.get('precision'))
data = SyntheticADE(int(cfg.data.samples_per_gpu), device=dev)
for _ in range(2):
    tr.step(data.next())
torch.cuda.synchronize()
if a.graph:
    only_captured = True
    with Log():
        ok = tr.enable_graph(data.next())
    print('enable_graph:', ok)
else:
    with Log():
        tr.step(data.next())
torch.cuda.synchronize()
tot = 0
for (name, tag, shp, site), n in sorted(groups.items(), key=lambda kv: -kv[1] * max(1, int(torch.tensor(kv[0][2]).prod()) if kv[0][2] else 1))[:70]:
    print(f'{n:4d} x {name:10s} {tag:28s} {str(shp):26s} {site}')
    tot += n
print('listed:', tot, 'of', sum(groups.values()))

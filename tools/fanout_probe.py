"""Which tensors of the student's autograd graph have more than one consumer (each costs one gradient-accumulation `add` launch per extra
consumer in the backward)?  Walks the graph of one config-2 KD step on the GPU and prints the producing nodes with fan-out > 1.
    python tools/fanout_probe.py [--config configs/kd/cfg2_segformer_b2_b0_cgd.py]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from segdistill_amd.config import Config  # noqa: E402
from segdistill_amd.engine import KDTrainer, SyntheticADE  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='configs/kd/cfg2_segformer_b2_b0_cgd.py')
a = ap.parse_args()
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(bench.ROOT, a.config))
torch.manual_seed(0)
model = bench.build_model(cfg, dev)
tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=1, precision=cfg.get('precision'))
data = SyntheticADE(int(cfg.data.samples_per_gpu), device=dev)
model.train()
with tr._autocast():
    out = model.train_step(data.next(), tr.optimizer)
loss = out['loss']
fan = collections.Counter()
seen, stack = set(), [loss.grad_fn]
while stack:
    fn = stack.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    for nxt, idx in fn.next_functions:
        if nxt is not None:
            fan[(nxt, idx)] += 1
            stack.append(nxt)
multi = [(k, v) for k, v in fan.items() if v > 1 and type(k[0]).__name__ != 'AccumulateGrad']
print(f'{len(seen)} nodes; {len(multi)} (node, output) pairs with more than one consumer, {sum(v - 1 for _, v in multi)} accumulation adds')
for (fn, idx), v in sorted(multi, key=lambda kv: -kv[1]):
    meta = fn.metadata if hasattr(fn, 'metadata') else {}
    shape = None
    try:
        shape = tuple(fn._input_metadata[0].shape) if hasattr(fn, '_input_metadata') else None
    except Exception:  # noqa: BLE001
        pass
    print(f'  x{v}  {type(fn).__name__}[{idx}]  {shape or ""}')

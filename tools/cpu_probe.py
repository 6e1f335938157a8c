"""Probe how the torch-CPU KD step scales with thread count on this host (bounded)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from segdistill_amd.config import Config
print('usable cores', bench._usable_cores(), 'cpu_count', os.cpu_count())
cfg = Config.fromfile(os.path.join(bench.ROOT, 'configs/kd/cfg2_segformer_b2_b0_cgd.py'))
for th in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '8,16,32,64').split(',')]:
    r = bench.cpu_baseline_leg(cfg, batch=2, budget_s=5.0, max_threads=th)
    print(th, r['value'], 'imgs/s', r['s_per_step'], 's/step', r['sample'][:60])

"""Is the KD step host-bound?  Compare host enqueue time per step with GPU time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from segdistill_amd.config import Config
from segdistill_amd.engine import KDTrainer, SyntheticADE
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(bench.ROOT, 'configs/kd/cfg2_segformer_b2_b0_cgd.py'))
torch.manual_seed(0)
model = bench.build_model(cfg, dev)
tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=1)
for B in (8, 2):
    data = SyntheticADE(B, device=dev)
    for _ in range(5):
        tr.step(data.next())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(data.next())
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'B={B}: host enqueue {t_host/10*1e3:.2f} ms/step, wall incl. GPU drain {t_all/10*1e3:.2f} ms/step')

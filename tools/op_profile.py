"""torch.profiler view of the KD step: which aten ops (with input shapes) the GPU time belongs to."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from segdistill_amd.config import Config
from segdistill_amd.engine import KDTrainer, SyntheticADE
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.environ.get('SEGDISTILL_PROFILE_CONFIG') or os.path.join(bench.ROOT, 'configs/kd/cfg2_segformer_b2_b0_cgd.py'))
torch.manual_seed(0)
model = bench.build_model(cfg, dev)
tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=1, precision=cfg.get('precision'))
data = SyntheticADE(8, device=dev)
for _ in range(4):
    tr.step(data.next())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        tr.step(data.next())
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda r: -r.self_device_time_total)
tot = sum(r.self_device_time_total for r in rows)
print(f'total device time {tot/3/1e3:.2f} ms/step')
for r in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 45]:
    print(f'{r.self_device_time_total/3/1e3:8.3f} ms  {r.count/3:6.1f}x  {r.key[:44]:44s} {str(r.input_shapes)[:110]}')
if len(sys.argv) > 2:   # second argument: comma-separated substrings -> op rows (total device time, incl. children) matching any of them
    pats = sys.argv[2].split(',')
    sel = [r for r in prof.key_averages(group_by_input_shape=True) if any(p in r.key for p in pats)]
    sel.sort(key=lambda r: -r.device_time_total)
    print(f'--- ops matching {pats}: total {sum(r.device_time_total for r in sel)/3/1e3:.3f} ms/step (nested ops counted more than once)')
    for r in sel[:60]:
        print(f'{r.device_time_total/3/1e3:8.3f} ms  {r.count/3:6.1f}x  {r.key[:30]:30s} {str(r.input_shapes)[:150]}')

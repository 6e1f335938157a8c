"""Which part of the KD step issues hipMemset (rocclr fillBuffer) commands?  Uses torch.profiler kernel names."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from segdistill_amd.config import Config
from segdistill_amd.engine import KDTrainer, SyntheticADE
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(bench.ROOT, 'configs/kd/cfg2_segformer_b2_b0_cgd.py'))
torch.manual_seed(0)
model = bench.build_model(cfg, dev)
model.teacher_on_side_stream = False
tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=1)
data = SyntheticADE(8, device=dev)
for _ in range(3):
    tr.step(data.next())
b = data.next()
def count(fn, label):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn(); torch.cuda.synchronize()
    rows = [r for r in prof.key_averages() if 'fillBuffer' in r.key or 'Memset' in r.key or 'memset' in r.key]
    print(label, {r.key[:40]: r.count for r in rows})
model.train()
count(lambda: model._teacher_forward(b['img'], None, None), 'teacher forward      :')
def student():
    tr.reducer.zero_grad()
    out = model.student(b['img'], None, return_loss=True, gt_semantic_seg=b['gt_semantic_seg'])
    loss = sum(v.mean() for k, v in out.items() if 'loss' in k)
    loss.backward()
count(student, 'student fwd+bwd (CE) :')
count(lambda: tr.optimizer.step(), 'optimizer step       :')

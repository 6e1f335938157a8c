"""Every BASELINE config (configs/kd/cfg1..cfg5) builds from its file and runs KD train steps on the GPU through the
product path: finite losses, the expected KD loss keys, gradients on the student AND on the 1x1 align projections,
teacher untouched.  These are the parity-test cases of BASELINE.json other than the benched configs[1]."""
import glob
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFGS = sorted(glob.glob(os.path.join(ROOT, 'configs', 'kd', 'cfg*.py')))


@pytest.mark.parametrize('path', CFGS, ids=[os.path.basename(p)[:4] for p in CFGS])
def test_config_runs_two_steps(path):
    import bench
    from segdistill_amd.config import Config
    from segdistill_amd.engine import KDTrainer, SyntheticADE
    dev = torch.device('cuda:0')
    cfg = Config.fromfile(path)
    torch.manual_seed(0)
    model = bench.build_model(cfg, dev)
    teacher_before = [p.detach().clone() for p in list(model.teacher.parameters())[:3]]
    tr = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), world=1, precision=cfg.get('precision'))
    data = SyntheticADE(2, size=(256, 256), device=dev, pool=2)
    n_kd = len(cfg.model.distillation)
    for it in range(2):
        out = tr.step(data.next())
        vals = tr.log_values()
        kd = {k: v for k, v in vals.items() if k.startswith('loss_')}
        assert len(kd) == n_kd, list(vals)
        assert all(math.isfinite(v) for v in vals.values()), vals
        assert all(v >= -1e-6 for v in kd.values()), kd   # a KL divergence
        assert vals['loss'] == pytest.approx(sum(v for k, v in vals.items() if 'loss' in k and k != 'loss'), rel=1e-4)
    aligns = list(model.distillation_loss.aligns.values())
    assert len(aligns) == sum(1 for d in cfg.model.distillation if d.get('channel_nums'))
    for a in aligns:
        assert a.weight.grad is not None and float(a.weight.grad.abs().sum()) > 0
    assert any(float(p.grad.abs().sum()) > 0 for p in model.student.backbone.parameters() if p.grad is not None)
    assert all(p.grad is None for p in model.teacher.parameters())
    for a, b in zip(teacher_before, list(model.teacher.parameters())[:3]):
        assert torch.equal(a, b)
    assert model.cnt == 2

"""MFMA 1x1 projection kernels vs a plain PyTorch reference of the same op (fp64 conv2d on CPU).
Tolerance: fp32 MFMA is an exact fmaf chain -> 2e-5 relative (max-norm scaled); bf16 storage -> 1e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cs, Ct, h, w
    (2, 128, 512, 16, 16),     # config-4 channel counts
    (1, 256, 768, 8, 8),       # config-5 channel counts (768 = 6 x 128)
    (2, 20, 37, 5, 7),         # nothing a multiple of the tile
    (1, 130, 129, 12, 11),     # tile edges + K tail
    (3, 16, 128, 32, 32),
]


def _ref(x, w, b, dy):
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    y = F.conv2d(x64, w64[:, :, None, None], b64)
    y.backward(dy.double())
    return y.detach(), x64.grad, w64.grad, b64.grad


def _err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_align1x1_fwd_bwd(case, dtype):
    from segdistill_amd.align import align1x1
    B, Cs, Ct, h, w = case
    g = torch.Generator().manual_seed(B * 1000 + Cs)
    x = torch.randn(B, Cs, h, w, generator=g).to(dtype)
    wt = torch.randn(Ct, Cs, generator=g) / Cs ** 0.5
    b = torch.randn(Ct, generator=g)
    dy = torch.randn(B, Ct, h, w, generator=g).to(dtype)
    y_ref, dx_ref, dw_ref, db_ref = _ref(x.float(), wt, b, dy.float())
    dev = torch.device('cuda:0')
    xg = x.to(dev).requires_grad_(True)
    wg = wt.to(dev).requires_grad_(True)
    bg = b.to(dev).requires_grad_(True)
    y = align1x1(xg, wg, bg)
    y.backward(dy.to(dev))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert y.dtype == dtype and xg.grad.dtype == dtype and wg.grad.dtype == torch.float32
    assert _err(y, y_ref) < tol
    assert _err(xg.grad, dx_ref) < tol
    assert _err(wg.grad, dw_ref) < (2e-5 if dtype == torch.float32 else 2e-5)  # fp32 accumulation of the SAME rounded inputs
    assert _err(bg.grad, db_ref) < 2e-5


def test_align1x1_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (cdna guide, MFMA section)."""
    from segdistill_amd.align import align1x1
    dev = torch.device('cuda:0')
    C = 64
    x = torch.arange(C * 6, dtype=torch.float32).reshape(1, C, 2, 3).to(dev)
    y = align1x1(x, torch.eye(C, device=dev), None)
    assert torch.equal(y, x)
    perm = torch.randperm(C)
    y = align1x1(x, torch.eye(C, device=dev)[perm], None)
    assert torch.equal(y, x[:, perm.to(dev)])


def test_feature_align_module_in_distillation_loss():
    """channel_nums=(Cs,Ct) builds a trainable FeatureAlign whose grads flow through the HIP criterion."""
    import segdistill_amd
    from segdistill_amd.distillation import DistillationLoss
    segdistill_amd.register_all()
    dev = torch.device('cuda:0')
    dl = DistillationLoss([dict(student_layer='a', teacher_layer='b', loss_name='KLDLoss', channel_nums=(8, 16),
                                loss_config=dict(alpha=2, tau=2, transform_config={'loss_type': 'channel', 'group_size': 4}))]).to(dev)
    s = torch.randn(2, 8, 16, 16, device=dev, requires_grad=True)
    t = torch.randn(2, 16, 16, 16, device=dev)
    out = dl({'a': s}, {'b': t}, torch.zeros(2, 1, 16, 16, dtype=torch.long, device=dev), 1)
    (k, v), = out.items()
    v.backward()
    al = dl.aligns['0']
    assert al.weight.grad is not None and al.weight.grad.abs().sum() > 0 and s.grad.abs().sum() > 0
    # reference math on the CPU in fp64
    from oracle import kd_ref
    w64, b64 = al.weight.detach().double().cpu(), al.bias.detach().double().cpu()
    s64 = s.detach().double().cpu().requires_grad_(True)
    proj = F.conv2d(s64, w64[:, :, None, None], b64)
    ref = kd_ref.eager_kld(proj, t.double().cpu(), alpha=2, tau=2, loss_type='channel', group_size=4)
    ref.backward()
    assert float(v) == pytest.approx(float(ref), rel=2e-5)
    assert _err(s.grad, s64.grad) < 1e-4

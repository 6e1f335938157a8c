"""csrc/pix_up.hip: the pixel-wise criterion (PDLoss; reference losses.py:47-49,101-102,108-112,115-128) with the bilinear up-sampling fused in,
against the fp64 oracle (oracle/kd_ref.full_kld with loss_type 'pixel': resize both, softmax over the classes of every pixel, KL, and the transposed
resize of the gradient) and against the unfused product path (resize.hip + pix_kl.hip).  Covers factors 2 / 4 / 8, tap widths that are not a multiple
of the workgroup size, class counts that are not a multiple of the backward's class block, bf16 storage, tau / alpha, an upstream factor, and -- at the
config-2 taps -- that nothing of label size is allocated any more."""
import numpy as np
import pytest
import torch

from oracle import kd_ref

pytestmark = pytest.mark.gpu

CASES = [  # B, C, h, w, F, tau, alpha
    (2, 6, 4, 4, 2, 1.0, 1.0),          # the golden G1 shape
    (1, 150, 9, 13, 4, 1.0, 1.0),       # config-2 classes, odd tap sizes
    (2, 7, 5, 70, 4, 2.0, 3.0),         # two waves per tap row, 7 classes (one class block of 6 + a block of 1)
    (1, 19, 6, 5, 8, 1.5, 2.0),         # factor 8 (row slices forward, class blocks of 2 backward)
    (1, 3, 1, 1, 2, 1.0, 1.0),          # a single tap: every output pixel clamps
    (1, 13, 17, 33, 2, 4.0, 1.0),       # factor 2, class block of 8 + 5
]


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300)))


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_pixel_criterion_matches_oracle(case, dtype):
    from segdistill_amd import ops
    B, C, h, w, F, tau, alpha = case
    g = torch.Generator().manual_seed(C * 100 + h)
    s = (2 * torch.randn(B, C, h, w, generator=g)).to(dtype)
    t = (2 * torch.randn(B, C, h, w, generator=g)).to(dtype)
    size = (F * h, F * w)
    ref = kd_ref.full_kld(s.double().numpy(), t.double().numpy(), alpha=alpha, tau=tau, out_size=size, loss_type='pixel')
    dev = torch.device('cuda:0')
    sg = s.to(dev).requires_grad_(True)
    assert ops.can_fuse_pixel_resize(sg, t.to(dev), size)
    loss = ops.pix_kl_up(sg, t.to(dev), size, tau=tau, alpha=alpha)
    up = 0.6
    (loss * up).backward()
    assert float(loss) == pytest.approx(ref['loss'], rel=3e-5, abs=1e-7)
    tol = 1e-4 if dtype == torch.float32 else 6e-3            # the bf16 gradient is STORED in bf16
    assert _rel_l2(sg.grad.float().cpu().numpy(), up * ref['grad_s']) < tol


def test_pdloss_module_takes_the_fused_path_and_equals_the_unfused_one():
    """PDLoss through the module at the config-2 tap shape [8,150,128,128] -> 512 x 512: the fused kernels against resize.hip + pix_kl.hip
    (fuse_resize = False), and no label-size tensor: the unfused path allocates 2 x 1.26 GB of up-sampled logits + 1.26 GB of gradient."""
    from segdistill_amd.distillation import PDLoss
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    s = (2 * torch.randn(8, 150, 128, 128, generator=g)).to(dev)
    t = (2 * torch.randn(8, 150, 128, 128, generator=g)).to(dev)
    gt = torch.zeros(8, 1, 512, 512, device=dev)
    res = {}
    for fused in (True, False):
        crit = PDLoss()
        crit.fuse_resize = fused
        sg = s.clone().requires_grad_(True)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = crit(sg, t, gt, 1)
        loss.backward()
        torch.cuda.synchronize()
        res[fused] = (float(loss), sg.grad.clone(), torch.cuda.max_memory_allocated() - base)
    assert res[True][0] == pytest.approx(res[False][0], rel=2e-5)
    assert _rel_l2(res[True][1].cpu().numpy(), res[False][1].cpu().numpy()) < 2e-4
    assert res[True][2] < 200e6 < 2.5e9 < res[False][2]       # fused: taps' gradient + two 8 MB log-partition maps; unfused: three label-size maps

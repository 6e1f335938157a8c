"""Parity at BASELINE.json's FULL sizes (config 2: taps [8,150,128,128] -> softmax at 512x512, g=8, tau=4, alpha=3),
where the CPU oracle would take minutes, through size-independent properties of the criterion:

 * the two HIP regimes agree: fused-upsample (R2) == ATen resize + streaming kernels (R1);
 * KL >= 0 per row, and exactly 0 (with zero gradient) when student == teacher;
 * softmax shift invariance: adding a constant to every logit of a row (= a group of channels) changes nothing;
 * every row's gradient sums to zero (the gradient of a function of softmax(S)), also after the transposed
   interpolation (bilinear weights sum to one per output);
 * linearity in alpha, and the channel shuffle is a pure re-grouping: permuting the channels of BOTH tensors with P and
   passing perm = P^-1 ... equals the unpermuted loss;
 * a checksum of checksums: sum of the per-row KLs == loss * rows / alpha.
The same operands at 1/16 of the size ARE checked against the oracle and the reference golden vectors in test_cgd_kl_gpu.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
B, C, h, w, Fk, g, tau, alpha = 8, 150, 128, 128, 4, 8, 4.0, 3.0
H, W = h * Fk, w * Fk


@pytest.fixture(scope='module')
def taps():
    dev = torch.device('cuda:0')
    gen = torch.Generator(device=dev).manual_seed(2024)
    s = 2 * torch.randn(B, C, h, w, device=dev, generator=gen)
    t = 2 * torch.randn(B, C, h, w, device=dev, generator=gen)
    return s, t


def _r2(s, t, **kw):
    from segdistill_amd import ops
    s = s.detach().clone().requires_grad_(True)
    args = dict(group_size=g, tau=tau, alpha=alpha)
    args.update(kw)
    loss, rows = ops.cgd_kl_up(s, t, (H, W), return_rows=True, **args)
    loss.backward()
    return loss.detach(), rows, s.grad


def test_full_size_r2_equals_r1_through_aten_resize(taps):
    from segdistill_amd import ops
    s, t = taps
    loss2, rows2, grad2 = _r2(s, t)
    s1 = s.detach().clone().requires_grad_(True)
    S = F.interpolate(s1, size=(H, W), mode='bilinear', align_corners=False)
    T = F.interpolate(t, size=(H, W), mode='bilinear', align_corners=False)
    loss1, rows1 = ops.cgd_kl(S, T, group_size=g, tau=tau, alpha=alpha, return_rows=True)
    loss1.backward()
    assert float(loss2) == pytest.approx(float(loss1), rel=1e-5)
    assert torch.allclose(rows2, rows1, rtol=1e-4, atol=1e-7)
    assert float((grad2 - s1.grad).norm() / s1.grad.norm()) < 1e-4
    rows = B * (-(-C // g))
    assert float(rows2.double().sum()) * alpha / rows == pytest.approx(float(loss2), rel=1e-5)   # checksum of checksums
    assert float(rows2.min()) >= -1e-6                                                          # KL >= 0


def test_full_size_identical_inputs_give_zero(taps):
    s, _ = taps
    loss, rows, grad = _r2(s, s.clone())
    assert abs(float(loss)) < 1e-6 and float(rows.abs().max()) < 1e-5
    assert float(grad.abs().max()) < 1e-9


def test_full_size_shift_invariance_and_zero_sum_gradient(taps):
    s, t = taps
    loss0, rows0, grad0 = _r2(s, t)
    G = -(-C // g)
    shift = torch.linspace(-3, 3, B * G, device=s.device).reshape(B, G, 1, 1, 1)          # one constant per softmax row
    pad = G * g - C
    s_sh = torch.cat([s, s.new_zeros(B, pad, h, w)], 1).reshape(B, G, g, h, w) + shift
    s_sh = s_sh.reshape(B, G * g, h, w)[:, :C].contiguous()
    loss1, rows1, grad1 = _r2(s_sh, t)
    assert float(loss1) == pytest.approx(float(loss0), rel=2e-5)
    assert float((grad1 - grad0).norm() / grad0.norm()) < 2e-4
    # per-row gradient sums vanish (compare with the row's absolute mass)
    gp = torch.cat([grad0, grad0.new_zeros(B, pad, h, w)], 1).reshape(B, G, -1).double()
    assert float((gp.sum(-1).abs() / gp.abs().sum(-1)).max()) < 1e-4


def test_full_size_alpha_linearity_and_shuffle_regrouping(taps):
    s, t = taps
    loss1, _, grad1 = _r2(s, t, alpha=1.0)
    loss3, _, grad3 = _r2(s, t, alpha=3.0)
    assert float(loss3) == pytest.approx(3 * float(loss1), rel=1e-6)
    assert float((grad3 - 3 * grad1).norm() / grad3.norm()) < 1e-6
    # shuffle = re-grouping: physically permuting the channels with `order` and reading them through the identity table
    # equals reading the ORIGINAL tensors through perm = order
    order = torch.randperm(C, generator=torch.Generator().manual_seed(7))
    dev_order = order.to(s.device)
    loss_a, _, grad_a = _r2(s, t, perm=dev_order.to(torch.int32))
    loss_b, _, grad_b = _r2(s[:, dev_order].contiguous(), t[:, dev_order].contiguous())
    assert float(loss_a) == pytest.approx(float(loss_b), rel=1e-6)
    assert float((grad_a[:, dev_order] - grad_b).norm() / grad_b.norm()) < 1e-6
    assert abs(float(loss_a) - float(loss3)) > 1e-7   # and it really is a different grouping


def test_full_size_fused_ce_equals_aten_chain():
    """The supervised loss at full size: fused up-sample + CE == F.interpolate + F.cross_entropy (fp32 on the GPU)."""
    from segdistill_amd.ce import fused_ce_up
    dev = torch.device('cuda:0')
    gen = torch.Generator(device=dev).manual_seed(5)
    x = (2 * torch.randn(B, C, h, w, device=dev, generator=gen)).requires_grad_(True)
    lab = torch.randint(0, C, (B, 1, H, W), device=dev, generator=gen)
    lab[torch.rand(B, 1, H, W, device=dev, generator=gen) < 0.05] = 255
    loss_pix, hits = fused_ce_up(x, lab, 255)
    loss_pix.mean().backward()
    gf = x.grad.clone()
    x.grad = None
    up = F.interpolate(x, size=(H, W), mode='bilinear', align_corners=False)
    ref = F.cross_entropy(up, lab.squeeze(1), reduction='none', ignore_index=255)
    ref.mean().backward()
    assert float((loss_pix - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert float((gf - x.grad).norm() / x.grad.norm()) < 1e-4
    assert abs(int(hits) - int((up.argmax(1) == lab.squeeze(1)).sum())) <= 16

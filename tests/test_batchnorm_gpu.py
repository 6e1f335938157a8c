"""csrc/batchnorm.hip (training BatchNorm + ReLU + channel dropout on tokens) against fp64 torch arithmetic: values, input and
parameter gradients, running statistics; the SegFormer head's fused tail against its generic module chain."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _reference(x, w, b, drop, relu, eps):
    """fp64, [B, N, C]: what BatchNorm2d(train) -> ReLU -> Dropout2d (with the given per-(image, channel) factors) computes."""
    x = x.double().requires_grad_(True)
    w = w.double().requires_grad_(True)
    b = b.double().requires_grad_(True)
    mean = x.mean(dim=(0, 1))
    var = x.var(dim=(0, 1), unbiased=False)
    pre = (x - mean) * torch.rsqrt(var + eps) * w + b
    z = pre.clamp_min(0) if relu else pre
    if drop is not None:
        z = z * drop.double()[:, None, :]
    return x, w, b, z, mean.detach(), var.detach(), pre.detach()


@pytest.mark.parametrize('B,N,C', [(2, 1000, 32), (3, 517, 256), (2, 333, 768), (1, 64, 1024), (2, 16384, 256)])
@pytest.mark.parametrize('relu,dropped', [(True, True), (False, False)])
def test_norm_act_matches_fp64(B, N, C, relu, dropped):
    from segdistill_amd import batchnorm as hip_bn
    torch.manual_seed(B * 1000 + C)
    dev = 'cuda:0'
    # |mean| >> std: the pivot matters.  (The largest case uses a milder offset: with 8 M elements a handful of fp32 pre-activations
    # round to the other side of the ReLU than their fp64 twins, which shows in the PARAMETER gradients at the 1e-4 level.)
    lo, hi = (-40, 60) if N < 10000 else (-4, 6)
    x = (torch.randn(B, N, C, device=dev) * 0.5 + torch.linspace(lo, hi, C, device=dev))
    norm = nn.BatchNorm2d(C).to(dev).train()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
        norm.running_mean.normal_()
        norm.running_var.uniform_(0.5, 2.0)
    rm0, rv0 = norm.running_mean.clone(), norm.running_var.clone()
    drop = None
    if dropped:
        drop = torch.empty(B, C, device=dev).bernoulli_(0.7).div_(0.7)
    gy = torch.randn(B, N, C, device=dev)
    xr, wr, br, zr, mean, var, pre = _reference(x, norm.weight.detach(), norm.bias.detach(), drop, relu, norm.eps)
    (zr * gy.double()).sum().backward()

    xt = x.clone().requires_grad_(True)
    assert hip_bn.supported(xt, norm)
    y = hip_bn.norm_act(xt, norm, relu=relu, drop=drop)
    (y * gy).sum().backward()
    assert torch.allclose(y.double(), zr, rtol=1e-4, atol=2e-4)

    def rel(a, b):
        return float((a.double() - b).norm() / b.norm().clamp_min(1e-30))
    # an fp32 pre-activation within rounding of zero may land on the other side of the ReLU than the fp64 one: leave those out
    safe = (pre.abs() > 1e-3) if relu else torch.ones_like(pre, dtype=torch.bool)
    assert float(safe.double().mean()) > 0.99
    assert rel(xt.grad * safe, xr.grad * safe) < 2e-4
    assert rel(norm.weight.grad, wr.grad) < 1e-4
    assert rel(norm.bias.grad, br.grad) < 1e-4
    n = B * N
    assert torch.allclose(norm.running_mean.double(), 0.9 * rm0.double() + 0.1 * mean, rtol=1e-5, atol=1e-5)
    assert torch.allclose(norm.running_var.double(), 0.9 * rv0.double() + 0.1 * var * n / (n - 1), rtol=1e-4, atol=1e-6)
    assert int(norm.num_batches_tracked) == 1


def test_norm_act_bf16_storage():
    from segdistill_amd import batchnorm as hip_bn
    torch.manual_seed(5)
    dev = 'cuda:0'
    B, N, C = 2, 2048, 256
    x = (torch.randn(B, N, C, device=dev) + 3.0).to(torch.bfloat16)
    norm = nn.BatchNorm2d(C).to(dev).train()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    drop = torch.empty(B, C, device=dev).bernoulli_(0.9).div_(0.9)
    gy = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    xr, wr, br, zr, _, _, pre = _reference(x.float(), norm.weight.detach(), norm.bias.detach(), drop, True, norm.eps)
    (zr * gy.double()).sum().backward()
    xt = x.clone().requires_grad_(True)
    y = hip_bn.norm_act(xt, norm, relu=True, drop=drop)
    assert y.dtype == torch.bfloat16
    (y.float() * gy.float()).sum().backward()
    assert float((y.double() - zr).abs().max()) < 3e-2
    safe = pre.abs() > 2e-2
    assert float(((xt.grad.double() - xr.grad) * safe).norm() / xr.grad.norm()) < 1e-2
    assert float((norm.weight.grad.double() - wr.grad).norm() / wr.grad.norm()) < 1e-2


def test_unsupported_inputs_are_refused():
    from segdistill_amd import batchnorm as hip_bn
    norm = nn.BatchNorm2d(30).cuda().train()
    assert not hip_bn.supported(torch.randn(2, 8, 30, device='cuda:0'), norm)            # C % 4
    norm = nn.BatchNorm2d(32).cuda()
    assert not hip_bn.supported(torch.randn(2, 8, 32, device='cuda:0'), norm.eval())     # eval mode: running statistics
    assert not hip_bn.supported(torch.randn(2, 8, 32), norm.train())                       # CPU tensor
    with pytest.raises(RuntimeError):
        hip_bn.local_stats(torch.randn(2, 8, 30, device='cuda:0'), 1e-5)


def _head():
    import segdistill_amd
    from segdistill_amd.builder import build_head
    segdistill_amd.register_all()
    torch.manual_seed(3)
    head = build_head(dict(type='SegFormerHead', in_channels=[32, 64, 160, 256], in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32], channels=128,
                           dropout_ratio=0.1, num_classes=19, norm_cfg=dict(type='SyncBN', requires_grad=True), align_corners=False,
                           decoder_params=dict(embed_dim=64), loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    return head.cuda().train()


def test_head_fused_tail_matches_module_chain():
    """Same head, same inputs: the fused tail (HIP passes) against the literal BatchNorm2d -> ReLU -> Dropout2d modules (forced by
    a hook on the activation).  Dropout is made the identity (p = 0) so that both draw the same -- no -- random numbers."""
    fused = _head()
    fused.dropout.p = 0.0
    plain = copy.deepcopy(fused)
    plain.linear_fuse.activate.register_forward_hook(lambda m, i, o: None)
    B = 2
    feats = [torch.randn(B, c, 64 // s, 64 // s, device='cuda:0').contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for c, s in zip([32, 64, 160, 256], [1, 2, 4, 8])]
    feats_p = [f.detach().clone().requires_grad_(True) for f in feats]
    g = torch.randn(B, 19, 64, 64, device='cuda:0')
    called = {}
    from segdistill_amd import batchnorm as hip_bn
    orig = hip_bn.norm_act

    def spy(*a, **k):
        called['n'] = called.get('n', 0) + 1
        return orig(*a, **k)
    hip_bn.norm_act = spy
    try:
        of = fused(feats)
        op = plain(feats_p)
    finally:
        hip_bn.norm_act = orig
    assert called.get('n') == 1                   # only the hook-free head took the fused tail
    assert torch.allclose(of, op, rtol=1e-4, atol=1e-4)
    (of * g).sum().backward()
    (op * g).sum().backward()
    for a, b in zip(feats, feats_p):
        assert float((a.grad - b.grad).norm() / b.grad.norm()) < 1e-4
    for (n1, p1), (_, p2) in zip(fused.named_parameters(), plain.named_parameters()):
        if p2.grad is None:
            assert p1.grad is None, n1
            continue
        # (the branch biases sit in front of a BatchNorm: their true gradient is zero and both sides hold rounding noise)
        assert float((p1.grad - p2.grad).norm()) < 2e-4 * float(p2.grad.norm()) + 1e-4, n1
    assert torch.allclose(fused.linear_fuse.norm.running_var, plain.linear_fuse.norm.running_var, rtol=1e-4, atol=1e-6)


def test_head_channel_dropout_statistics():
    """With p > 0 the fused tail drops whole (image, channel) planes and rescales the rest by 1/(1-p), like Dropout2d."""
    head = _head()
    head.dropout.p = 0.5
    seen = {}
    head.linear_pred.register_forward_pre_hook(lambda m, inp: seen.setdefault('x', inp[0].detach()))
    feats = [torch.randn(4, c, 64 // s, 64 // s, device='cuda:0').contiguous(memory_format=torch.channels_last) for c, s in zip([32, 64, 160, 256], [1, 2, 4, 8])]
    head(feats)
    x = seen['x']                                  # [B, E, H, W] after norm -> relu -> dropout
    plane_zero = (x.abs().amax(dim=(2, 3)) == 0)
    frac = float(plane_zero.float().mean())
    assert 0.3 < frac < 0.7

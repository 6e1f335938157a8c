"""csrc/ppm_pool.hip: every adaptive average pool of a Pyramid Pooling Module in one pass each way, against F.adaptive_avg_pool2d in fp64
(forward AND the gradient of a weighted sum of all scales), incl. sizes the scales do not divide (overlapping bins), a single scale, bf16,
run-to-run bit identity of the backward (ATen's is float atomics), and the PPM module with / without the kernel."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, C, h, w, scales
    (2, 5, 64, 64, (1, 2, 3, 6)),       # PSPNet at 512 x 512 (stride 8)
    (1, 3, 16, 16, (1, 2, 3, 6)),       # UPerHead on the coarsest Swin level
    (2, 4, 33, 47, (1, 2, 3, 6)),       # nothing divides: overlapping bins, scalar loads
    (1, 2, 7, 9, (6,)),                 # bins of one or two rows
    (1, 6, 96, 96, (2, 4, 8)),
    (3, 2, 12, 20, (1, 5, 7, 8)),
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_ppm_pool_matches_adaptive_avg_pool(case, dtype):
    from segdistill_amd import ppm
    B, C, h, w, scales = case
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(h * 131 + w)
    x = torch.randn(B, C, h, w, generator=g).to(dtype).to(dev)
    assert ppm.supported(x, list(scales))
    xr = x.clone().requires_grad_(True)
    outs = ppm.ppm_pool(xr, scales)
    x64 = x.double().requires_grad_(True)
    refs = [F.adaptive_avg_pool2d(x64, s) for s in scales]
    ws = [torch.randn(o.shape, generator=g).to(dev) for o in outs]
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    for o, r in zip(outs, refs):
        assert o.shape == r.shape and o.dtype == dtype
        assert float((o.double() - r).abs().max()) <= tol * max(1.0, float(r.abs().max()))
    sum((o.float() * wk).sum() for o, wk in zip(outs, ws)).backward()
    sum((r * wk.to(dtype).double()).sum() for r, wk in zip(refs, ws)).backward()
    err = float((xr.grad.double() - x64.grad).norm() / x64.grad.norm())
    assert err < (1e-6 if dtype == torch.float32 else 1e-2), err


def test_ppm_pool_backward_is_deterministic_and_the_module_uses_it():
    """PPM.forward takes the kernel when it can (and nn.AdaptiveAvgPool2d otherwise) with the same values; the gather backward is run-to-run
    bit-identical.  The 1x1 ConvModules of the branches are replaced by identities here: this test is about the pooling + resize wiring, and
    the library convolution / BatchNorm kernels on 1x1 ... 6x6 maps are exercised by the PSPNet train-step fixtures."""
    import torch.nn as nn
    from segdistill_amd import ppm
    from segdistill_amd.decode_heads.psp_head import PPM
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    mod = PPM((1, 2, 3, 6), 16, 16, conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), align_corners=False).to(dev)
    for branch in mod:
        branch[1] = nn.Identity()
    x = torch.randn(2, 16, 64, 64, device=dev)
    calls = {'n': 0}
    real = ppm.ppm_pool

    def counting(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    ppm.ppm_pool = counting
    grads = []
    try:
        for enabled in (True, True, False):
            ppm.ENABLED = enabled
            xr = x.clone().requires_grad_(True)
            outs = mod(xr)
            sum(o.square().sum() for o in outs).backward()
            torch.cuda.synchronize()
            grads.append((xr.grad.clone(), [o.detach().clone() for o in outs]))
    finally:
        ppm.ENABLED = True
        ppm.ppm_pool = real
    assert calls['n'] == 2                                             # the two enabled passes went through the kernel, the third did not
    assert torch.equal(grads[0][0], grads[1][0])                       # gather backward: run-to-run identical
    for a, b in zip(grads[0][1], grads[2][1]):
        assert float((a - b).abs().max()) < 1e-5 * max(1.0, float(b.abs().max()))
    assert float((grads[0][0] - grads[2][0]).norm() / grads[2][0].norm()) < 1e-5

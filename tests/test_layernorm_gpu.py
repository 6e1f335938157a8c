"""Token-major LayerNorm HIP kernels vs a plain PyTorch reference of the same op (F.layer_norm in fp64 on the CPU).
fp32: 2e-5 max-norm relative on y / dx, 1e-4 on dgamma / dbeta (sums over up to 131072 rows); bf16 storage: 1e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('shape', [(2, 4096, 32), (1, 1000, 64), (3, 77, 160), (2, 300, 256), (2, 100, 320), (1, 65, 512), (1, 33, 768),
                                   (1, 9, 1024), (8, 16384, 32), (1, 5, 128)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_layernorm_fwd_bwd(shape, dtype):
    from segdistill_amd.layernorm import HipLayerNorm
    B, N, C = shape
    g = torch.Generator().manual_seed(N + C)
    x = (2 * torch.randn(B, N, C, generator=g) + 0.5).to(dtype)
    w = 1 + 0.1 * torch.randn(C, generator=g)
    b = 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(B, N, C, generator=g).to(dtype)
    x64 = x.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    F.layer_norm(x64, (C,), w64, b64, 1e-6).backward(dy.double())
    ref = F.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-6)
    dev = torch.device('cuda:0')
    ln = HipLayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        ln.weight.copy_(w)
        ln.bias.copy_(b)
    xg = x.to(dev).requires_grad_(True)
    y = ln(xg)
    y.backward(dy.to(dev))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert y.dtype == dtype and _err(y, ref) < tol
    assert _err(xg.grad, x64.grad) < tol
    assert _err(ln.weight.grad, w64.grad) < (1e-4 if dtype == torch.float32 else 1e-2)
    assert _err(ln.bias.grad, b64.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


def test_hip_layernorm_is_a_layernorm_and_falls_back_on_cpu():
    import torch.nn as nn
    from segdistill_amd.layernorm import HipLayerNorm
    ln = HipLayerNorm(48)
    assert isinstance(ln, nn.LayerNorm) and sorted(ln.state_dict()) == ['bias', 'weight']
    x = torch.randn(3, 7, 48)
    assert torch.allclose(ln(x), F.layer_norm(x, (48,), ln.weight, ln.bias, ln.eps))

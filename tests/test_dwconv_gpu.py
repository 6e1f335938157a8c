"""Token-major depth-wise 3x3 HIP kernels vs a plain PyTorch reference of the same op
(F.conv2d groups=C in fp64 on the CPU).  fp32: 1e-5 (max-norm relative); bf16 storage: 1e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, C
    (2, 16, 16, 128),
    (1, 7, 5, 8),
    (2, 33, 70, 64),     # W not a multiple of the strip / segment, several segments per row
    (1, 16, 16, 2048),   # widest MiT hidden size (B2..B5 stage 4)
    (3, 1, 9, 16),
    (1, 9, 1, 32),
    (2, 64, 64, 256),
]


def _err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dwconv_tokens_fwd_bwd(case, dtype):
    from segdistill_amd.dwconv import dwconv3x3_tokens, supported
    B, H, W, C = case
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, H * W, C, generator=g).to(dtype)
    w = torch.randn(C, 1, 3, 3, generator=g) / 3
    b = torch.randn(C, generator=g)
    dy = torch.randn(B, H * W, C, generator=g).to(dtype)
    x64 = x.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    y_ref = F.conv2d(x64.transpose(1, 2).reshape(B, C, H, W), w64, b64, padding=1, groups=C).flatten(2).transpose(1, 2)
    y_ref.backward(dy.double())
    dev = torch.device('cuda:0')
    xg = x.to(dev).requires_grad_(True)
    wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    assert supported(xg, wg)
    y = dwconv3x3_tokens(xg, wg, bg, H, W)
    y.backward(dy.to(dev))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert _err(y, y_ref) < tol
    assert _err(xg.grad, x64.grad) < tol
    assert _err(wg.grad, w64.grad) < 2e-5
    assert _err(bg.grad, b64.grad) < 2e-5


def test_mit_block_uses_hip_dwconv_and_matches_torch_path():
    """The MiT Mix-FFN takes the HIP depth-wise path on the GPU; result equals the NCHW torch path."""
    import segdistill_amd
    from segdistill_amd.backbones.mit import MixFFN
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    ffn = MixFFN(32, 128).to(dev)
    x = torch.randn(2, 24 * 20, 32, device=dev, requires_grad=True)
    y = ffn(x, (24, 20))
    y.sum().backward()
    gx, gw = x.grad.clone(), ffn.dwconv.dwconv.weight.grad.clone()
    x.grad = None
    ffn.zero_grad()
    h = ffn.fc1(x)
    b, n, c = h.shape
    h = ffn.dwconv.dwconv(h.transpose(1, 2).reshape(b, c, 24, 20)).flatten(2).transpose(1, 2)
    y2 = ffn.fc2(ffn.act(h))
    y2.sum().backward()
    assert _err(y, y2) < 1e-5 and _err(gx, x.grad) < 1e-4 and _err(gw, ffn.dwconv.dwconv.weight.grad) < 1e-4


def test_dwconv_gelu_inference_kernel():
    from segdistill_amd.dwconv import dwconv3x3_gelu_tokens_inference
    B, H, W, C = 2, 20, 18, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H * W, C, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) / 3
    b = torch.randn(C, generator=g)
    ref = F.gelu(F.conv2d(x.double().transpose(1, 2).reshape(B, C, H, W), w.double(), b.double(), padding=1, groups=C)).flatten(2).transpose(1, 2)
    y = dwconv3x3_gelu_tokens_inference(x.cuda(), w.cuda(), b.cuda(), H, W)
    assert _err(y, ref) < 1e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('deferred_scope', [False, True])
def test_dwconv_gelu_training_form(dtype, deferred_scope):
    """conv + exact GELU in one pass with autograd (the Mix-FFN of the student), immediate and deferred weight-gradient combine."""
    import contextlib
    from segdistill_amd import deferred
    from segdistill_amd.dwconv import dwconv3x3_gelu_tokens
    B, H, W, C = 2, 33, 18, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H * W, C, generator=g).to(dtype)
    w = torch.randn(C, 1, 3, 3, generator=g) / 3
    b = torch.randn(C, generator=g)
    dy = torch.randn(B, H * W, C, generator=g).to(dtype)
    x64 = x.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    y_ref = F.gelu(F.conv2d(x64.transpose(1, 2).reshape(B, C, H, W), w64, b64, padding=1, groups=C)).flatten(2).transpose(1, 2)
    y_ref.backward(dy.double())
    dev = torch.device('cuda:0')
    xg = x.to(dev).requires_grad_(True)
    wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = dwconv3x3_gelu_tokens(xg, wg, bg, H, W)
    with (deferred.scope() if deferred_scope else contextlib.nullcontext()):
        y.backward(dy.to(dev))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert _err(y, y_ref) < tol
    assert _err(xg.grad, x64.grad) < tol
    assert wg.grad.shape == (C, 1, 3, 3) and wg.grad.is_contiguous()
    assert _err(wg.grad, w64.grad) < (2e-5 if dtype == torch.float32 else 2e-2)
    assert _err(bg.grad, b64.grad) < (2e-5 if dtype == torch.float32 else 2e-2)


def test_gelu_error_bound():
    """The branch-free erf of the fused GELU epilogue (csrc/dwconv.hip::gelu_erf) against the f64 GELU: a centre-tap-only filter turns the
    kernel into an element-wise GELU.  Bound 6e-7 absolute over [-12, 12] (the f32 library form 0.5 v (1 + erff(v / sqrt 2)) sits at
    4.5e-7), and the negative tail keeps its RELATIVE accuracy instead of flushing to zero where 1 + erf cancels."""
    from segdistill_amd.dwconv import dwconv3x3_gelu_tokens_inference
    dev = torch.device('cuda:0')
    C = 64
    v = torch.linspace(-12, 12, 64 * 64 * C, dtype=torch.float64)
    x = v.float().reshape(1, 64 * 64, C).to(dev)
    w = torch.zeros(C, 1, 3, 3, device=dev)
    w[:, 0, 1, 1] = 1.0
    y = dwconv3x3_gelu_tokens_inference(x, w, None, 64, 64).double().cpu().reshape(-1)
    xv = x.double().cpu().reshape(-1)
    ref = 0.5 * xv * torch.special.erfc(-xv / 2 ** 0.5)
    assert float((y - ref).abs().max()) < 6e-7
    tail = (xv < -3) & (xv > -9)
    assert float(((y - ref).abs() / ref.abs())[tail].max()) < 2e-2     # erfc(6.4) = 1e-19: the library form returns exactly 0 here
    mid = xv.abs() < 3
    assert float(((y - ref).abs() / ref.abs().clamp_min(1e-6))[mid].max()) < 1e-5

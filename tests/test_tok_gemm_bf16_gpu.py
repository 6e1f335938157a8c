"""csrc/tok_gemm_bf16.hip: Y = X . W^T + bias for bf16 operands (autocast's F.linear of every MiT / head Linear under bf16 storage, BASELINE config 5;
reference mix_transformer.py:24-27,48-55,75-84,107-133) against fp64 on the same bf16-rounded operands: the result must be the correctly
rounded bf16 of the exact value up to fp32 accumulation error -- within one bf16 ulp of the fp64 result everywhere, and closer than the library
GEMM's own result on average is not required, only the same bound.  Covers every tile / ring variant, ragged token counts, K = 64 (one k-step),
long K, both bias dtypes and no bias, and the module-level dispatch under autocast (forward + backward through _TokenLinear)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # tokens, in, out
    (8192, 320, 320), (8192, 320, 1280), (8192, 1280, 320), (2048, 320, 640), (2048, 1280, 320),
    (32768, 128, 128), (131072, 64, 64), (131072, 64, 256), (2048, 512, 2048), (2048, 2048, 512),
    (1000, 64, 64), (129, 192, 128), (1, 64, 64), (4097, 4096, 64),
]


def _ref(x, w, b):
    y = x.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    return y


def _check(y, ref, K):
    err = (y.double() - ref).abs()
    # one rounding to bf16 (half an ulp = 2^-9 relative) + fp32 accumulation of K products of bf16 values
    tol = ref.abs() * 2.0 ** -8 + 1e-6 * K ** 0.5 * 4
    bad = (err > tol).sum().item()
    assert bad == 0, f'{bad} of {err.numel()} outside one bf16 ulp; worst {float((err / (ref.abs() + 1e-3)).max()):.3e}'


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('bias', ['bf16', 'f32', None])
def test_forward_matches_fp64(shape, bias):
    from segdistill_amd import linear
    T, K, N = shape
    if bias != 'bf16' and T > 9000:
        pytest.skip('bias variants at the small shapes only')
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(T + K + N)
    x = torch.randn(T, K, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    b = None if bias is None else torch.randn(N, generator=g).to(dev).to(torch.bfloat16 if bias == 'bf16' else torch.float32)
    from segdistill_amd import _lib
    assert _lib.lib().sd_linear_bf16_fwd_supported(T, K, N)
    y = torch.empty(T, N, dtype=torch.bfloat16, device=dev)      # straight through the C ABI: the Python dispatch sends some of these shapes to the library
    _lib.check(_lib.lib().sd_linear_bf16_fwd(x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(), 0 if b is None else linear._DT[b.dtype],
                                             y.data_ptr(), T, K, N, torch.cuda.current_stream().cuda_stream), 'sd_linear_bf16_fwd')
    assert y.dtype == torch.bfloat16 and y.shape == (T, N)
    _check(y, _ref(x, w, b), K)


@pytest.mark.parametrize('variant', range(5))
def test_every_tile_variant(variant):
    """Every (tile, wave layout, ring depth) instantiation, forced through the A/B tunable, on shapes its channel tile divides (a forced variant that
    does not divide N falls back to the dispatch -- still checked)."""
    from segdistill_amd import _lib, linear
    L = _lib.lib()
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(variant)
    try:
        assert L.sd_set_tunable(b'tok_gemm_bf16_variant', variant) == 0
        for T, K, N in [(700, 320, 640), (2048, 64, 1280), (300, 704, 320), (513, 128, 128)]:
            x = torch.randn(T, K, generator=g).to(dev).bfloat16()
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
            b = torch.randn(N, generator=g).to(dev).bfloat16()
            assert linear.bf16_tok_gemm_ok(T, K, N)
            _check(linear.linear_fwd_bf16(x, w, b), _ref(x, w, b), K)
    finally:
        L.sd_set_tunable(b'tok_gemm_bf16_variant', -1)


def test_unsupported_shapes_fall_back_to_the_library():
    from segdistill_amd import linear
    dev = torch.device('cuda:0')
    x = torch.randn(512, 96, device=dev).bfloat16()            # in % 64 != 0
    w = torch.randn(150, 96, device=dev).bfloat16()            # out % 64 != 0
    assert not linear.bf16_tok_gemm_ok(512, 96, 150)
    y = linear.linear_fwd_bf16(x, w, None)
    assert torch.equal(y, torch.nn.functional.linear(x, w))


def test_token_linear_under_autocast_uses_the_kernel_and_matches_f_linear():
    """Module-level: _TokenLinear forward under bf16 autocast (the student's path) -- same values as F.linear up to the accumulation order, and the
    backward (dX, dW, db) unchanged in meaning."""
    from segdistill_amd import linear
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn(4, 2048, 320, generator=g).to(dev).requires_grad_(True)
    lin = torch.nn.Linear(320, 640).to(dev)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y = linear.token_linear(x, lin.weight, lin.bias, defer_ok=False)
        yr = torch.nn.functional.linear(x, lin.weight, lin.bias)
    assert y.dtype == torch.bfloat16
    assert (y.float() - yr.float()).abs().max() <= 2.0 ** -7 * yr.float().abs().max()
    gy = torch.randn_like(y)
    y.backward(gy)
    gx, gw, gb = x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()
    x.grad = None
    lin.zero_grad()
    yr.backward(gy)
    assert torch.allclose(gx, x.grad, rtol=2e-2, atol=2e-2)
    assert torch.allclose(gw, lin.weight.grad, rtol=2e-2, atol=5e-2)
    assert torch.allclose(gb, lin.bias.grad, rtol=2e-2, atol=5e-2)

"""fc2(GELU(dwconv3x3(h) + b)) of a frozen Mix-FFN as one kernel (csrc/mixffn_tail.hip; reference mix_transformer.py:20-55) against the fp64 form
of the same three operations and against the two-kernel route it replaces; borders, both channel widths, the dispatch in MixFFN.forward."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ref64(h, conv, fc2, H, W):
    B, N, C = h.shape
    x = h.double().transpose(1, 2).reshape(B, C, H, W)
    x = F.conv2d(x, conv.weight.double(), conv.bias.double(), padding=1, groups=C)
    x = F.gelu(x).flatten(2).transpose(1, 2)
    return F.linear(x, fc2.weight.double(), fc2.bias.double())


@pytest.mark.parametrize('B,H,W,dim', [(2, 128, 128, 64), (1, 64, 64, 128), (3, 8, 16, 64), (1, 16, 48, 128), (2, 24, 32, 64)])
def test_tail_matches_fp64_and_the_two_kernel_route(B, H, W, dim):
    from segdistill_amd import mixffn
    from segdistill_amd.backbones.mit import MixFFN
    torch.manual_seed(H + dim)
    m = MixFFN(dim, 4 * dim).to(DEV).eval()
    with torch.no_grad():
        m.dwconv.dwconv.weight.mul_(3.0)
        m.dwconv.dwconv.bias.normal_()
        m.fc2.bias.normal_()
        h = torch.randn(B, H * W, 4 * dim, device=DEV) * 1.5
        conv = m.dwconv.dwconv
        assert mixffn._lib.lib().sd_mixffn_tail_supported(H, W, 4 * dim, dim)
        y = mixffn.tail(h, conv, m.fc2, (H, W))
        ref = _ref64(h, conv, m.fc2, H, W)
        from segdistill_amd import dwconv as hip_dw
        from segdistill_amd.linear import call_linear
        two = call_linear(m.fc2, hip_dw.dwconv3x3_gelu_tokens_inference(h, conv.weight, conv.bias, H, W))
    scale = float(ref.abs().max())
    e_one = float((y.double() - ref).abs().max()) / scale
    e_two = float((two.double() - ref).abs().max()) / scale
    assert e_one < 2e-6, (e_one, e_two)
    assert e_one < 4 * e_two + 1e-7, (e_one, e_two)


@pytest.mark.parametrize('B,H,W,dim', [(2, 128, 128, 64), (4, 64, 64, 128), (1, 8, 16, 64)])
def test_tail_under_bf16_autocast_matches_the_two_kernel_route(B, H, W, dim):
    """bf16 activations (config 5's teacher): the activated map and fc2's weight are rounded to bf16 exactly where the two-kernel route rounds them,
    so the two differ by accumulation order and the final rounding only."""
    from segdistill_amd import dwconv as hip_dw, mixffn
    from segdistill_amd.backbones.mit import MixFFN
    from segdistill_amd.linear import call_linear
    torch.manual_seed(H + dim)
    m = MixFFN(dim, 4 * dim).to(DEV).eval()
    conv = m.dwconv.dwconv
    with torch.no_grad():
        conv.weight.mul_(3.0)
        conv.bias.normal_()
        m.fc2.bias.normal_()
        h = (torch.randn(B, H * W, 4 * dim, device=DEV) * 1.5).bfloat16()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            assert mixffn.usable(h, conv, m.fc2, (H, W)) or B * H * W < mixffn._MIN_TOKENS
            y = mixffn.tail(h, conv, m.fc2, (H, W))
            g = hip_dw.dwconv3x3_gelu_tokens_inference(h, conv.weight, conv.bias, H, W)
            two = call_linear(m.fc2, g)
        assert y.dtype == torch.bfloat16 and two.dtype == torch.bfloat16
        ref = torch.nn.functional.linear(g.double(), m.fc2.weight.bfloat16().double(), m.fc2.bias.double())     # the same rounded operands in fp64
    s_ = float(ref.abs().max())
    e_one = float((y.double() - ref).abs().max()) / s_
    e_two = float((two.double() - ref).abs().max()) / s_
    assert e_one < 6e-3 and e_one < 1.5 * e_two + 1e-3, (e_one, e_two)


def test_mixffn_forward_takes_the_fused_tail_only_when_frozen(monkeypatch):
    from segdistill_amd import mixffn
    from segdistill_amd.backbones.mit import MixFFN
    torch.manual_seed(0)
    m = MixFFN(64, 256).to(DEV).eval()
    for p in m.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 128 * 128, 64, device=DEV)
    calls = []
    real = mixffn.tail
    monkeypatch.setattr(mixffn, 'tail', lambda *a: (calls.append(1), real(*a))[1])
    with torch.no_grad():
        a = m(x, (128, 128))
        assert calls == [1]
        monkeypatch.setattr(mixffn, '_ENABLED', False)
        b = m(x, (128, 128))
        assert calls == [1]
        monkeypatch.setattr(mixffn, '_ENABLED', True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            c = m(x.bfloat16(), (128, 128))          # bf16 activations under autocast (config 5's teacher): the bf16 form of the same kernel
        assert calls == [1, 1] and c.dtype == torch.bfloat16
        m.fc2.register_forward_hook(lambda mod, i, o: None)
        m(x, (128, 128))
        assert calls == [1, 1]
    m2 = MixFFN(64, 256).to(DEV)
    m2(x, (128, 128))             # autograd on: the training route
    assert calls == [1, 1]
    assert float((a - b).abs().max()) < 2e-6 * float(b.abs().max()) + 1e-6


def test_unsupported_shapes_are_refused():
    from segdistill_amd import _lib
    L = _lib.lib()
    assert not L.sd_mixffn_tail_supported(12, 16, 256, 64)
    assert not L.sd_mixffn_tail_supported(16, 24, 256, 64)
    assert not L.sd_mixffn_tail_supported(16, 16, 256, 32)
    assert not L.sd_mixffn_tail_supported(16, 16, 200, 64)
    h = torch.zeros(1, 12 * 16, 256, device=DEV)
    rc = L.sd_mixffn_tail(h.data_ptr(), h.data_ptr(), h.data_ptr(), h.data_ptr(), h.data_ptr(), h.data_ptr(), 0, 1, 12, 16, 256, 64, None)
    assert rc == -6

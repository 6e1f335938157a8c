"""The MiT patch-embedding convolutions as window gather + token GEMM (csrc/patch_embed.hip, segdistill_amd/patch_embed.py) against
nn.Conv2d in fp64 -- reference mix_transformer.py:185-215 (OverlapPatchEmbed: Conv2d(k = 7, s = 4, p = 3) / (3, 2, 1), flatten(2).transpose(1, 2)).
Forward, input gradient (the transposed gather), filter and bias gradients (through the backward's grouped weight-gradient launch), the gather
itself against F.unfold through the C ABI, ragged sizes, bf16 storage, and run-to-run bit identity (what MIOpen's atomics-based filter
gradients do not give)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _conv(cin, cout, k, s, p, seed=0):
    torch.manual_seed(seed)
    c = nn.Conv2d(cin, cout, k, s, p).to(DEV)
    c.weight.data = c.weight.data.contiguous(memory_format=torch.channels_last)      # as backbones/mit.py keeps it
    return c


def _input(B, cin, H, W, channels_last, seed=1):
    g = torch.Generator(device='cpu').manual_seed(seed)
    if channels_last:      # a channels-last VIEW of a token map, as the stage inputs are
        tok = torch.randn(B, H * W, cin, generator=g).to(DEV)
        return tok, tok.reshape(B, H, W, cin).permute(0, 3, 1, 2)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV)
    return x, x


CASES = [  # B, cin, cout, H, W, k, s, p, channels-last input
    (2, 3, 32, 64, 64, 7, 4, 3, False),        # stage 1 on the NCHW image (K = 147 -> 152 padded columns, scalar gather)
    (2, 3, 64, 36, 52, 7, 4, 3, False),        # ragged: 36 x 52 -> 9 x 13
    (2, 32, 64, 32, 32, 3, 2, 1, True),        # stage 2 of B0
    (1, 64, 160, 15, 21, 3, 2, 1, True),       # odd sizes: 15 x 21 -> 8 x 11
    (2, 160, 256, 8, 8, 3, 2, 1, True),        # stage 4 of B0
    (1, 8, 16, 12, 12, 5, 3, 2, True),         # not a MiT shape: k = 5, s = 3
]


@pytest.mark.parametrize('case', CASES)
def test_forward_and_gradients_match_conv2d_in_fp64(case):
    from segdistill_amd import deferred, patch_embed
    B, cin, cout, H, W, k, s, p, cl = case
    conv = _conv(cin, cout, k, s, p)
    leaf, x = _input(B, cin, H, W, cl)
    leaf.requires_grad_(True)
    x = leaf.reshape(B, H, W, cin).permute(0, 3, 1, 2) if cl else leaf
    assert patch_embed.supported(x, conv)
    y, hw = patch_embed.patch_embed_tokens(x, conv)
    ref_conv = nn.Conv2d(cin, cout, k, s, p).to(DEV).double()
    ref_conv.load_state_dict({n: v.double() for n, v in conv.state_dict().items()})
    xr = x.detach().double().contiguous().requires_grad_(True)
    yr = ref_conv(xr)
    assert hw == tuple(yr.shape[2:]) and y.shape == (B, hw[0] * hw[1], cout)
    yr_tok = yr.flatten(2).transpose(1, 2)
    assert float((y.double() - yr_tok).abs().max()) <= 2e-5 * (float(yr_tok.abs().max()) + 1.0)
    g = torch.randn(y.shape, generator=torch.Generator(device='cpu').manual_seed(5)).to(DEV)
    with deferred.scope():
        y.backward(g)
    yr_tok.backward(g.double())
    torch.cuda.synchronize()
    gw, gb = conv.weight.grad.double(), conv.bias.grad.double()
    assert float((gw - ref_conv.weight.grad).abs().max()) <= 3e-5 * (float(ref_conv.weight.grad.abs().max()) + 1.0)
    assert float((gb - ref_conv.bias.grad).abs().max()) <= 3e-5 * (float(ref_conv.bias.grad.abs().max()) + 1.0)
    gx = leaf.grad.double().reshape(B, H, W, cin).permute(0, 3, 1, 2) if cl else leaf.grad.double()
    assert float((gx - xr.grad).abs().max()) <= 3e-5 * (float(xr.grad.abs().max()) + 1.0)


def test_gather_equals_unfold_through_the_c_abi():
    from segdistill_amd import _lib
    L = _lib.lib()
    for (B, cin, H, W, k, s, p) in [(2, 3, 33, 47, 7, 4, 3), (2, 16, 20, 24, 3, 2, 1)]:
        x = torch.randn(B, cin, H, W, device=DEV)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        K = k * k * cin
        Kp = -(-K // 8) * 8
        col = torch.full((B, Ho * Wo, Kp), 7.0, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        assert L.sd_im2col_tokens(x.data_ptr(), col.data_ptr(), 0, B, H, W, cin, *x.stride(), k, s, p, Ho, Wo, Kp, st) == 0
        # F.unfold orders a window (ci, ky, kx); ours is (ky, kx, ci)
        ref = F.unfold(x, k, padding=p, stride=s).reshape(B, cin, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B, Ho * Wo, K)
        torch.cuda.synchronize()
        assert torch.equal(col[..., :K], ref) and float(col[..., K:].abs().sum()) == 0.0
        # argument checks: wrong output size, unknown dtype
        assert L.sd_im2col_tokens(x.data_ptr(), col.data_ptr(), 0, B, H, W, cin, *x.stride(), k, s, p, Ho + 1, Wo, Kp, st) == -2
        assert L.sd_im2col_tokens(x.data_ptr(), col.data_ptr(), 9, B, H, W, cin, *x.stride(), k, s, p, Ho, Wo, Kp, st) == -3
        assert L.sd_col2im_tokens(col.data_ptr(), None, 0, B, H, W, cin, k, s, p, Ho, Wo, Kp, st) == -1


def test_bf16_storage_under_autocast_matches_fp64_on_the_rounded_operands():
    from segdistill_amd import deferred, patch_embed
    B, cin, cout, H, W, k, s, p = 2, 64, 128, 32, 32, 3, 2, 1
    conv = _conv(cin, cout, k, s, p)
    tok = torch.randn(B, H * W, cin, device=DEV).bfloat16().requires_grad_(True)
    x = tok.reshape(B, H, W, cin).permute(0, 3, 1, 2)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y, hw = patch_embed.patch_embed_tokens(x, conv)
    assert y.dtype == torch.bfloat16
    wr = conv.weight.detach().bfloat16().double()
    br = conv.bias.detach().bfloat16().double()
    xr = x.detach().double().contiguous().requires_grad_(True)
    wr.requires_grad_(True)
    yr = F.conv2d(xr, wr, br, s, p).flatten(2).transpose(1, 2)
    assert float((y.double() - yr).abs().max()) <= 1e-2 * (float(yr.abs().max()) + 1.0)
    g = torch.randn(y.shape, device=DEV).bfloat16()
    with deferred.scope():
        y.backward(g)
    yr.backward(g.double())
    torch.cuda.synchronize()
    assert conv.weight.grad.dtype == torch.float32
    assert float((conv.weight.grad.double() - wr.grad).abs().max()) <= 1e-2 * (float(wr.grad.abs().max()) + 1.0)
    gx = tok.grad.double().reshape(B, H, W, cin).permute(0, 3, 1, 2)
    assert float((gx - xr.grad).abs().max()) <= 2e-2 * (float(xr.grad.abs().max()) + 1.0)


def test_gradients_are_run_to_run_bit_identical():
    from segdistill_amd import deferred, patch_embed
    outs = []
    for _ in range(2):
        conv = _conv(32, 64, 3, 2, 1, seed=3)
        tok = torch.randn(4, 64 * 64, 32, generator=torch.Generator(device='cpu').manual_seed(9)).to(DEV).requires_grad_(True)
        x = tok.reshape(4, 64, 64, 32).permute(0, 3, 1, 2)
        y, _ = patch_embed.patch_embed_tokens(x, conv)
        with deferred.scope():
            (y * y).sum().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), tok.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_hooked_projection_and_switch_fall_back_to_the_module_call(monkeypatch):
    from segdistill_amd import patch_embed
    conv = _conv(32, 64, 3, 2, 1)
    _, x = _input(1, 32, 16, 16, True)
    assert patch_embed.supported(x, conv)
    h = conv.register_forward_hook(lambda m, i, o: None)
    assert not patch_embed.supported(x, conv)          # a tap on `backbone.patch_embedN.proj` must see a module call (opts.py:48-56)
    h.remove()
    monkeypatch.setattr(patch_embed, '_ENABLED', False)
    assert not patch_embed.supported(x, conv)

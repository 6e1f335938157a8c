"""csrc/window_attn.hip (Swin window attention of a frozen network, one kernel over the qkv Linear's output) against the reference's arithmetic
(mmseg/models/backbones/swin_transformer.py:119-153) restated in fp64, and the Swin block / backbone with the kernel against the same modules
on the framework's attention."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qkv, bias, mask, heads, scale):
    """swin_transformer.py:127-148 in fp64: qkv [bw, n, 3C]; bias [h, n, n]; mask [nW, n, n] or None."""
    bw, n, c3 = qkv.shape
    c = c3 // 3
    q, k, v = qkv.double().reshape(bw, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    attn = (q * scale) @ k.transpose(-2, -1) + bias.double().unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(bw // nw, nw, heads, n, n) + mask.double().unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    return (attn.softmax(-1) @ v).transpose(1, 2).reshape(bw, n, c)


@pytest.mark.parametrize('bw,nw,heads', [(8, 0, 4), (24, 4, 4), (50, 25, 16), (9, 9, 32), (722, 361, 4), (3, 1, 1), (1, 0, 3)])
def test_window_attn_kernel_matches_fp64(bw, nw, heads):
    from segdistill_amd import window_attn
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(bw * 7 + heads)
    n, d = 49, 32
    c = heads * d
    qkv = torch.randn(bw, n, 3 * c, device=dev, generator=g) * 1.5
    bias = torch.randn(heads, n, n, device=dev, generator=g)                       # NOT symmetric: catches a transposition slip
    mask = None
    if nw:
        mask = torch.where(torch.rand(nw, n, n, device=dev, generator=g) < 0.3, -100.0, 0.0)   # not symmetric either
        mask[:, torch.arange(n), torch.arange(n)] = 0.0
    scale = d ** -0.5
    if mask is not None and nw > 1:
        mask[0] = 0.0                        # an all-zero mask table must read as "no mask" through its flag
    ref = _ref(qkv, bias, mask, heads, scale)
    bias_p, _ = window_attn.pack_tables(bias, float('-inf'))
    mask_p, flags = window_attn.pack_tables(mask, 0.0, True) if mask is not None else (None, None)
    if flags is not None and nw > 1:
        assert int(flags[0]) == 0 and int(flags[1:].min()) == 1
    out_p = window_attn.forward_packed(qkv, bias_p, mask_p, flags, heads, scale)
    err = float((out_p.double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    # the framework's fp32 attention on the same operands is no closer to fp64
    q, k, v = qkv.reshape(bw, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    add = bias.unsqueeze(0) if mask is None else (bias.unsqueeze(0) + mask.unsqueeze(1)).repeat(bw // nw, 1, 1, 1)
    lib = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=add, scale=scale).transpose(1, 2).reshape(bw, n, c)
    err_lib = float((lib.double() - ref).abs().max() / ref.abs().max())
    assert err < 4 * err_lib + 1e-7, (err, err_lib)


def test_window_attn_rejects_what_it_does_not_cover():
    from segdistill_amd import _lib
    L = _lib.lib()
    assert L.sd_window_attn_supported(49, 32) == 1 and L.sd_window_attn_supported(144, 32) == 0 and L.sd_window_attn_supported(49, 64) == 0
    assert L.sd_window_attn_packed_floats() == 4096
    dev = torch.device('cuda:0')
    x = torch.zeros(2, 49, 3 * 32, device=dev)
    b = torch.zeros(1, 4096, device=dev)
    f = torch.zeros(2, dtype=torch.int32, device=dev)
    o = torch.empty(2, 49, 32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    args = (x.data_ptr(), b.data_ptr(), None, None, o.data_ptr())
    assert L.sd_window_attn_fwd_packed(*args, 1, 2, 0, 1, 49, 32, 1.0, st) == -3       # bf16: SD_E_DTYPE
    assert L.sd_window_attn_fwd_packed(*args, 0, 2, 0, 1, 64, 32, 1.0, st) == -6       # N: SD_E_UNSUPPORTED
    assert L.sd_window_attn_fwd_packed(x.data_ptr(), b.data_ptr(), b.data_ptr(), f.data_ptr(), o.data_ptr(), 0, 3, 2, 1, 49, 32, 1.0, st) == -2  # 3 windows, 2 masks
    assert L.sd_window_attn_fwd_packed(x.data_ptr(), b.data_ptr(), b.data_ptr(), None, o.data_ptr(), 0, 2, 2, 1, 49, 32, 1.0, st) == -2        # mask without flags
    assert L.sd_window_attn_fwd_packed(None, b.data_ptr(), None, None, o.data_ptr(), 0, 2, 0, 1, 49, 32, 1.0, st) == -1
    assert L.sd_window_attn_pack(x.data_ptr(), b.data_ptr(), None, 1, 144, 0.0, st) == -6


@pytest.mark.parametrize('hw', [(28, 28), (30, 26)])      # whole windows; padded bottom / right
def test_swin_stage_with_the_kernel_matches_the_framework_attention(hw, monkeypatch):
    """a BasicLayer of a frozen Swin (unshifted + shifted block, pad, cyclic shift, mask): kernel path vs the same modules on SDPA."""
    from segdistill_amd import window_attn
    from segdistill_amd.backbones.swin import BasicLayer
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    layer = BasicLayer(dim=96, depth=2, num_heads=3, window_size=7).to(dev).eval()
    for blk in layer.blocks:
        torch.nn.init.normal_(blk.attn.relative_position_bias_table, std=0.5)
    for p in layer.parameters():
        p.requires_grad_(False)
    h, w = hw
    x = torch.randn(2, h * w, 96, device=dev)
    calls = []
    real = window_attn.forward_packed
    monkeypatch.setattr(window_attn, 'forward_packed', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        y_k = layer(x, h, w)[0]
    assert len(calls) == 2
    monkeypatch.setattr(window_attn, 'ENABLED', False)
    with torch.no_grad():
        y_f = layer(x, h, w)[0]
    assert len(calls) == 2
    assert float((y_k - y_f).abs().max() / y_f.abs().max()) < 2e-5
    # with a graph to build the kernel path is not taken
    monkeypatch.setattr(window_attn, 'ENABLED', True)
    for p in layer.parameters():
        p.requires_grad_(True)
    layer(x, h, w)[0].sum().backward()
    assert len(calls) == 2 and layer.blocks[0].attn.qkv.weight.grad is not None

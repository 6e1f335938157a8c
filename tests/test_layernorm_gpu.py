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


@pytest.mark.parametrize('shape', [(2, 4096, 32), (3, 77, 160), (2, 100, 320), (1, 65, 512), (8, 16384, 64), (4, 5, 128)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('scaled', [False, True])
@pytest.mark.parametrize('use_xsum', [True, False])
def test_add_layernorm_fwd_bwd(shape, dtype, scaled, use_xsum):
    """Residual form: (xsum, y) = (x + s*res, LN(x + s*res)) and its gradients, against fp64 torch on the CPU.  `use_xsum=False`
    is the last half-block of a stage, whose residual stream has no other consumer (no gradient arrives for xsum)."""
    from segdistill_amd.layernorm import HipLayerNorm, add_layernorm, add_layernorm_supported
    B, N, C = shape
    g = torch.Generator().manual_seed(N + C)
    x = (2 * torch.randn(B, N, C, generator=g) + 0.5).to(dtype)
    r = torch.randn(B, N, C, generator=g).to(dtype)
    s = (torch.rand(B, generator=g) > 0.3).float() / 0.7 if scaled else None
    w = 1 + 0.1 * torch.randn(C, generator=g)
    b = 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(B, N, C, generator=g).to(dtype)
    dxs = torch.randn(B, N, C, generator=g).to(dtype)
    x64, r64 = x.double().requires_grad_(True), r.double().requires_grad_(True)
    w64, b64 = w.double().requires_grad_(True), b.double().requires_grad_(True)
    xs64 = x64 + (r64 if s is None else r64 * s.double().view(B, 1, 1))
    if dtype == torch.bfloat16:   # the kernel normalises the STORED (bf16-rounded) sum: mirror it, keeping the graph
        xs64 = xs64 + (xs64.detach().to(dtype).double() - xs64.detach())
    y64 = F.layer_norm(xs64, (C,), w64, b64, 1e-6)
    ((y64 * dy.double()).sum() + ((xs64 * dxs.double()).sum() if use_xsum else 0.)).backward()
    dev = torch.device('cuda:0')
    ln = HipLayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        ln.weight.copy_(w)
        ln.bias.copy_(b)
    xg, rg = x.to(dev).requires_grad_(True), r.to(dev).requires_grad_(True)
    assert add_layernorm_supported(xg, rg, ln)
    xsum, y = add_layernorm(xg, rg, ln, None if s is None else s.to(dev))
    ((y.float() * dy.to(dev).float()).sum() + ((xsum.float() * dxs.to(dev).float()).sum() if use_xsum else 0.)).backward()
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert xsum.dtype == dtype and y.dtype == dtype
    assert _err(xsum, xs64.detach()) < (1e-6 if dtype == torch.float32 else 4e-3)
    assert _err(y, y64.detach()) < tol
    assert _err(xg.grad, x64.grad) < tol
    assert _err(rg.grad, r64.grad) < tol
    assert _err(ln.weight.grad, w64.grad) < (1e-4 if dtype == torch.float32 else 1.5e-2)
    assert _err(ln.bias.grad, b64.grad) < (1e-4 if dtype == torch.float32 else 1.5e-2)


def test_fused_stage_loop_equals_block_by_block():
    """MiT stage with the residual adds fused into the LayerNorms == the literal per-block form (which a forward hook on any
    block forces), forward and parameter gradients, drop-path off."""
    import segdistill_amd
    from segdistill_amd import layernorm
    from segdistill_amd.builder import build_backbone
    segdistill_amd.register_all()
    torch.manual_seed(0)
    net = build_backbone(dict(type='mit_b0')).cuda()
    net.reset_drop_path(0.)
    net.train()
    img = torch.randn(2, 3, 128, 128, device='cuda')
    calls, real, real_p = [], layernorm._AddLayerNormFn.apply, layernorm._AddLayerNormPatchFn.apply
    layernorm._AddLayerNormFn.apply = lambda *a: (calls.append(1), real(*a))[1]
    layernorm._AddLayerNormPatchFn.apply = lambda *a: (calls.append(2), real_p(*a))[1]     # round 3: the form that also feeds an SR conv
    try:
        outs = net(img)
    finally:
        layernorm._AddLayerNormFn.apply, layernorm._AddLayerNormPatchFn.apply = real, real_p
    assert len(calls) == 2 * sum(net.depths)          # every residual add of every block went through a fused kernel
    assert calls.count(2) == 3                        # block 2 of stages 1-3: its norm1 output also goes out in patch order
    # A loss WITH a gradient: sum of mean(o^2) would not do -- the stage outputs are LayerNorm outputs with gamma = 1, beta = 0, whose mean
    # square is 1 whatever the input, so every gradient would be rounding noise and the comparison below would hold only while the library
    # GEMMs happen to round identically in both passes (they stop doing so once earlier tests have given hipBLASLt a workspace).
    probes = [torch.randn_like(o) for o in outs]
    sum((o.float() * w).mean() for o, w in zip(outs, probes)).backward()
    g_fused = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.zero_grad(set_to_none=True)
    handles = [blk.register_forward_hook(lambda m, i, o: None) for s in range(1, 5) for blk in getattr(net, f'block{s}')]
    outs2 = net(img)
    sum((o.float() * w).mean() for o, w in zip(outs2, probes)).backward()
    for h in handles:
        h.remove()
    for a, b2 in zip(outs, outs2):
        assert _err(a, b2) < 1e-5
    bad = [(n, _err(g_fused[n], p.grad)) for n, p in net.named_parameters() if not _err(g_fused[n], p.grad) < 2e-4]
    assert not bad, (len(bad), bad[:12])


# ---- round 3: LayerNorm output also in the SR conv's patch order, backward gathering the patch-order gradient ---------------------------------
def _gather_patches(t, H, W, r):
    b, n, c = t.shape
    return t.reshape(b, H // r, r, W // r, r, c).permute(0, 1, 3, 2, 4, 5).reshape(b, (H // r) * (W // r), r * r * c)


@pytest.mark.parametrize('case', [(2, 16, 16, 32, 8), (1, 32, 64, 64, 4), (3, 8, 8, 160, 2), (2, 128, 128, 32, 8), (1, 16, 32, 320, 2)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('residual', [False, True])
def test_layernorm_patch_form_matches_norm_then_gather(case, dtype, residual):
    """sd_add_layernorm_patch_fwd / _bwd against fp64 LayerNorm followed by the r x r patch gather (mix_transformer.py:75-84,121-124), with
    gradients arriving in BOTH orders (token order from the q Linear, patch order from the SR conv) and, in the residual form, on the sum."""
    from segdistill_amd.layernorm import HipLayerNorm, add_layernorm_patches, layernorm_patches, patch_supported
    B, H, W, C, r = case
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(H * W + C + r)
    x = torch.randn(B, H * W, C, device=dev, generator=g).to(dtype)
    res = torch.randn(B, H * W, C, device=dev, generator=g).to(dtype)
    sc = torch.tensor([1.0, 0.0, 1.25][:B], device=dev) if residual else None
    norm = HipLayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(C, device=dev, generator=g))
        norm.bias.copy_(0.2 * torch.randn(C, device=dev, generator=g))
    gy = torch.randn(B, H * W, C, device=dev, generator=g).to(dtype)
    gp = torch.randn(B, (H // r) * (W // r), r * r * C, device=dev, generator=g).to(dtype)
    gs = torch.randn(B, H * W, C, device=dev, generator=g).to(dtype)
    assert patch_supported(x, (H, W), r)
    xg, rg = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    if residual:
        xsum, y, yp = add_layernorm_patches(xg, rg, norm, (H, W), r, sc)
        torch.autograd.backward([xsum, y, yp], [gs, gy, gp])
    else:
        y, yp = layernorm_patches(xg, norm, (H, W), r)
        torch.autograd.backward([y, yp], [gy, gp])
    assert torch.equal(yp, _gather_patches(y, H, W, r))                    # the patch output is the token output, re-ordered
    # fp64 composition
    x64, r64 = x.double().requires_grad_(True), res.double().requires_grad_(True)
    w64, b64 = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    s64 = x64 + (sc.double().view(-1, 1, 1) * r64 if sc is not None else r64) if residual else x64
    if residual and dtype == torch.bfloat16:
        s64 = s64 + (s64.detach().to(torch.bfloat16).double() - s64.detach())      # the stored sum is what is normalised
    y64 = torch.nn.functional.layer_norm(s64, (C,), w64, b64, 1e-6)
    outs, grads = [y64, _gather_patches(y64, H, W, r)], [gy.double(), gp.double()]
    if residual:
        outs, grads = [s64] + outs, [gs.double()] + grads
    torch.autograd.backward(outs, grads)
    tol = 2e-5 if dtype == torch.float32 else 3e-2

    def rel(a, b):
        return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(y, y64.detach()) < tol
    assert rel(xg.grad, x64.grad) < tol
    if residual:
        assert rel(xsum, s64.detach()) < tol and rel(rg.grad, r64.grad) < tol
    assert rel(norm.weight.grad, w64.grad) < (1e-4 if dtype == torch.float32 else 5e-2) and rel(norm.bias.grad, b64.grad) < (1e-4 if dtype == torch.float32 else 5e-2)


def test_mit_stage_with_patch_fused_norms_matches_the_gather_copy_path():
    """MiT-B0 forward + backward with the SR paths fed by the patch-form LayerNorms against the same network with the round-2 gather copies."""
    import copy
    import segdistill_amd
    from segdistill_amd.backbones import mit
    from segdistill_amd.builder import BACKBONES
    from segdistill_amd.registry import build_from_cfg
    segdistill_amd.register_all()
    torch.manual_seed(3)
    a = build_from_cfg(dict(type='mit_b0'), BACKBONES).cuda().train()
    a.reset_drop_path(0.)
    b = copy.deepcopy(a)
    img = torch.randn(2, 3, 256, 256, device='cuda:0')
    ups = None
    outs = {}
    for tag, net, flag in (('fused', a, True), ('copy', b, False)):
        mit._LN_PATCHES = flag
        try:
            feats = net(img)
            if ups is None:
                ups = [torch.randn_like(f) for f in feats]
            torch.autograd.backward(list(feats), ups)
        finally:
            mit._LN_PATCHES = True
        outs[tag] = [f.detach() for f in feats]
    for fa, fb in zip(outs['fused'], outs['copy']):
        assert float((fa - fb).abs().max()) <= 1e-5 * float(fb.abs().max())
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert pa.grad is not None and pb.grad is not None, n
        assert float((pa.grad - pb.grad).norm()) <= 2e-5 * float(pb.grad.norm()) + 1e-8, n


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C', [96, 128, 512, 1024])
def test_layernorm_map_is_gather_add_layer_norm(dtype, C):
    """csrc/layernorm.hip::ln_map_fwd (the Swin blocks of a frozen network): output rows gathered through x_map (value L = a zero row), a
    residual gathered through res_map, the sum written back in x's order -- against index_select / add / F.layer_norm in fp64."""
    from segdistill_amd.layernorm import HipLayerNorm, layernorm_map, map_supported
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(C)
    B, L, n_out = 3, 37, 49
    norm = HipLayerNorm(C).to(dev)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, device=dev, generator=g))
        norm.bias.copy_(torch.randn(C, device=dev, generator=g))
    x = torch.randn(B, L, C, device=dev, generator=g).to(dtype)
    pend = torch.randn(B, L, C, device=dev, generator=g).to(dtype)
    perm = torch.randperm(n_out, device=dev, generator=g)
    x_map = torch.full((n_out,), L, dtype=torch.int32, device=dev)
    x_map[perm[:L]] = torch.arange(L, dtype=torch.int32, device=dev)          # every row of x exactly once, 12 padding rows
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    with torch.no_grad():
        assert map_supported(x, norm)
        # (a) norm(x + pending) gathered into n_out rows with zero padding rows; the sum comes back in token order
        xs, y = layernorm_map(x, norm, res=pend, x_map=x_map)
        s64 = x.double() + pend.double()
        assert float((xs.double() - s64).abs().max()) <= tol * float(s64.abs().max())
        ref = torch.nn.functional.layer_norm(xs.double(), (C,), norm.weight.double(), norm.bias.double(), norm.eps)
        ref = torch.cat([ref, ref.new_zeros(B, 1, C)], 1).index_select(1, x_map.long())
        assert float((y.double() - ref).abs().max()) <= tol * float(ref.abs().max())
        assert float(y[:, x_map == L].abs().max()) == 0.0
        # without a residual nothing is written back
        xs0, y0 = layernorm_map(x, norm, x_map=x_map)
        assert xs0 is None and y0.shape == (B, n_out, C)
        # (b) x + res gathered through the inverse map, then the norm, in token order
        win = torch.randn(B, n_out, C, device=dev, generator=g).to(dtype)
        inv = torch.empty(L, dtype=torch.int32, device=dev)
        inv[x_map[x_map < L].long()] = torch.nonzero(x_map < L).flatten().int()
        xs2, y2 = layernorm_map(x, norm, res=win, res_map=inv)
        s64 = x.double() + win.double().index_select(1, inv.long())
        assert float((xs2.double() - s64).abs().max()) <= tol * float(s64.abs().max())
        ref2 = torch.nn.functional.layer_norm(xs2.double(), (C,), norm.weight.double(), norm.bias.double(), norm.eps)
        assert float((y2.double() - ref2).abs().max()) <= tol * float(ref2.abs().max())
    assert not map_supported(x.requires_grad_(True), norm)                     # a graph to build: the autograd modules

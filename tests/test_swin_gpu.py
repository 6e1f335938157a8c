"""Frozen Swin (the config-4 teacher) fast paths of round 3 -- pad + cyclic shift + window partition as ONE cached row gather (window reverse +
un-shift + un-pad as another), and the relative-position bias + shift mask + batch repeat cached per block -- against the literal module chain:
pure data movement and a cached copy of the same tensor, so the feature pyramids must be identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('size', [(224, 224), (160, 200)])       # windows that divide the map, and maps that need padding on both axes
def test_frozen_swin_gather_path_equals_the_module_chain(size):
    import segdistill_amd
    from segdistill_amd.backbones import swin
    segdistill_amd.register_all()
    torch.manual_seed(0)
    net = swin.SwinTransformer(embed_dim=32, depths=(2, 2, 2), num_heads=(2, 4, 8), window_size=7, out_indices=(0, 1, 2)).cuda().eval()
    for p in net.parameters():
        p.requires_grad = False
    img = torch.randn(2, 3, *size, device='cuda:0')
    outs = {}
    for flag in (True, False):
        swin._GATHER_WINDOWS = flag
        try:
            with torch.no_grad():
                outs[flag] = [f.clone() for f in net(img)]
                again = net(img)                       # second call: the cached bias tensors are used
            for a, b in zip(outs[flag], again):
                assert torch.equal(a, b)
        finally:
            swin._GATHER_WINDOWS = True
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    # round 4: the fused block form (norm1 + window partition in one kernel, window reverse + residual + norm2 in another, the MLP residual
    # folded into the next block's first kernel) against the separate kernels: the same arithmetic on the same values
    fused = outs[True]
    swin._FUSE_NORMS = False
    try:
        with torch.no_grad():
            plain = net(img)
    finally:
        swin._FUSE_NORMS = True
    for a, b in zip(fused, plain):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    blk = net.layers[0].blocks[0]
    blk.H = blk.W = 8
    assert blk.fusable(torch.zeros(1, 64, 32, device='cuda:0'))
    # a tapped block is called as a module and hands its hook the complete block output (the KD Extractor's contract)
    seen = []
    h = net.layers[1].blocks[1].register_forward_hook(lambda m, i, o: seen.append(o.detach().clone()))
    try:
        with torch.no_grad():
            hooked = net(img)
    finally:
        h.remove()
    assert len(seen) == 1 and not net.layers[1].blocks[1]._forward_hooks
    for a, b in zip(fused, hooked):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    with torch.no_grad():                              # the hooked block's output is what the next stage consumed
        swin._FUSE_NORMS = False
        try:
            seen2 = []
            h = net.layers[1].blocks[1].register_forward_hook(lambda m, i, o: seen2.append(o.detach().clone()))
            net(img)
            h.remove()
        finally:
            swin._FUSE_NORMS = True
    assert float((seen[0] - seen2[0]).abs().max()) <= 2e-6 * float(seen2[0].abs().max())
    # a trainable Swin keeps the autograd-friendly chain (index_select's backward would be an atomic scatter)
    for p in net.parameters():
        p.requires_grad = True
    x = img.clone().requires_grad_(True)
    feats = net.train()(x)
    sum(f.mean() for f in feats).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


def test_partially_frozen_swin_keeps_the_gradients_of_its_trainable_parts():
    """frozen patch embedding and norms, TRAINABLE attention (qkv, proj, relative-position table) and MLP, grad mode on, input without
    a graph -- the no-graph block kernels (forward_fused, the packed window attention) must step aside: the gradients of the trainable parameters
    equal those of the literal module chain (fast paths switched off)."""
    import segdistill_amd
    from segdistill_amd.backbones import swin
    segdistill_amd.register_all()
    torch.manual_seed(0)
    net = swin.SwinTransformer(embed_dim=32, depths=(2, 2), num_heads=(2, 4), window_size=7, out_indices=(0, 1)).cuda().eval()
    for name, p in net.named_parameters():
        p.requires_grad = ('attn.' in name) or ('mlp.' in name)
    assert any(p.requires_grad for p in net.parameters()) and not all(p.requires_grad for p in net.parameters())
    img = torch.randn(2, 3, 112, 120, device='cuda:0')
    grads = {}
    for fast in (True, False):
        swin._GATHER_WINDOWS = swin._FUSE_NORMS = fast
        try:
            for p in net.parameters():
                p.grad = None
            sum(f.square().mean() for f in net(img)).backward()
            grads[fast] = {n: p.grad.clone() for n, p in net.named_parameters() if p.requires_grad}
        finally:
            swin._GATHER_WINDOWS = swin._FUSE_NORMS = True
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) > 0
    for n in grads[False]:
        a, b = grads[True][n], grads[False][n]
        assert a is not None and float(b.abs().max()) > 0, n
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-9, n

"""SegFormer-head up-sample-and-sum kernels vs ATen (F.interpolate bilinear + adds, fp64 CPU), and the whole head
on the GPU (HIP path) vs the same head on the CPU (ATen path) with identical weights."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('shape', [(2, 16, 16, 32), (1, 32, 24, 64), (2, 8, 8, 256), (1, 128, 128, 128)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_upsum_fwd_bwd(shape, dtype):
    from segdistill_amd.headfuse import supported, upsum
    B, H, W, E = shape
    g = torch.Generator().manual_seed(H * W + E)
    sizes = [(H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    zs = [torch.randn(B, h * w, E, generator=g).to(dtype) for (h, w) in sizes]
    bias = torch.randn(E, generator=g)
    dy = torch.randn(B, H * W, E, generator=g).to(dtype)
    z64 = [z.double().requires_grad_(True) for z in zs]
    b64 = bias.double().requires_grad_(True)
    tot = None
    for z, (h, w) in zip(z64, sizes):
        m = z.reshape(B, h, w, E).permute(0, 3, 1, 2)
        if (h, w) != (H, W):
            m = F.interpolate(m, size=(H, W), mode='bilinear', align_corners=False)
        tot = m if tot is None else tot + m
    ref = (tot + b64.view(1, -1, 1, 1)).permute(0, 2, 3, 1).reshape(B, H * W, E)
    ref.backward(dy.double())
    dev = torch.device('cuda:0')
    zg = [z.to(dev).requires_grad_(True) for z in zs]
    bg = bias.to(dev).requires_grad_(True)
    assert supported(zg, sizes)
    y = upsum(zg[0], zg[1], zg[2], zg[3], bg, sizes)
    y.backward(dy.to(dev))
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert _err(y, ref) < tol
    for a, r in zip(zg, z64):
        assert _err(a.grad, r.grad) < tol
    assert _err(bg.grad, b64.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('train', [True, False])
def test_segformer_head_gpu_equals_cpu(train):
    import segdistill_amd
    from segdistill_amd.builder import build_head
    segdistill_amd.register_all()
    torch.manual_seed(0)
    head = build_head(dict(type='SegFormerHead', in_channels=[32, 64, 160, 256], in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32],
                           channels=128, dropout_ratio=1e-12, num_classes=150, norm_cfg=dict(type='BN'), align_corners=False,
                           decoder_params=dict(embed_dim=256), loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    head.train(train)
    feats = [torch.randn(2, c, s, s) for c, s in ((32, 32), (64, 16), (160, 8), (256, 4))]
    out_cpu = head(feats)
    import copy
    hg = copy.deepcopy(head).cuda()
    out_gpu = hg([f.cuda() for f in feats])
    assert _err(out_gpu, out_cpu) < 1e-4


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_frozen_head_folds_norm_and_relu_into_the_sum(dtype):
    """Eval + no_grad (the frozen teacher): up-sample-sum, BatchNorm(running stats) and ReLU run as ONE kernel; result equals
    the same head with the separate BatchNorm / ReLU modules (forced by a hook on the norm) and the CPU reference."""
    import copy
    import segdistill_amd
    from segdistill_amd import headfuse
    from segdistill_amd.builder import build_head
    segdistill_amd.register_all()
    torch.manual_seed(1)
    head = build_head(dict(type='SegFormerHead', in_channels=[32, 64, 160, 256], in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32],
                           channels=128, dropout_ratio=0.1, num_classes=150, norm_cfg=dict(type='BN'), align_corners=False,
                           decoder_params=dict(embed_dim=256), loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    bn = head.linear_fuse.bn
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    head.eval()
    feats = [torch.randn(2, c, s, s) for c, s in ((32, 32), (64, 16), (160, 8), (256, 4))]
    with torch.no_grad():
        out_cpu = head(feats)
    hg = copy.deepcopy(head).cuda()
    gf = [f.cuda().to(dtype) for f in feats]
    calls, real = [], headfuse.upsum_affine_inference
    headfuse.upsum_affine_inference = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            out_fused = hg(gf)
            h = hg.linear_fuse.bn.register_forward_hook(lambda m, i, o: None)      # an observer on the norm: literal path
            out_plain = hg(gf)
            h.remove()
            with torch.enable_grad():                                               # gradients wanted: never folded
                hg(gf)
    finally:
        headfuse.upsum_affine_inference = real
    assert len(calls) == 1
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert _err(out_fused, out_plain) < tol and _err(out_fused, out_cpu) < (1e-4 if dtype == torch.float32 else 5e-2)


@pytest.mark.parametrize('shape', [(2, 128, 128, 64), (1, 32, 128, 64), (8, 128, 128, 256), (3, 16, 128, 192)])
def test_upsum_bwd3_band_kernel_matches_the_two_pass_form_and_fp64(shape):
    """Round 3's one-pass sd_upsum_bwd3 (bands of 16 output rows, per-tap-row accumulators in registers, no global partials; W = 128, fp32)
    against the two-pass form it replaces (tunable upsum_bwd_band = 0) and against the fp64 transpose of the three bilinear up-samplings;
    band borders (first / last band: clamped taps), a single band (H = 16), channel counts of 2, 3, 6, 8 slices of 32."""
    from segdistill_amd import _lib
    from segdistill_amd.ops import _stream_ptr
    B, H, W, E = shape
    dev = torch.device('cuda:0')
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(H + E)
    dy = torch.randn(B, H * W, E, device=dev, generator=g)
    outs = {}
    for mode in (1, 0):
        _lib.set_tunable('upsum_bwd_band', mode)
        try:
            dz = [torch.full((B, (H // f) * (W // f), E), float('nan'), device=dev) for f in (2, 4, 8)]
            wsb = L.sd_upsum_bwd3_workspace_bytes(B, H, W, E)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            _lib.check(L.sd_upsum_bwd3(dy.data_ptr(), dz[0].data_ptr(), dz[1].data_ptr(), dz[2].data_ptr(), 0, B, H, W, E, ws.data_ptr(), wsb,
                                       _stream_ptr()), 'sd_upsum_bwd3')
            outs[mode] = dz
        finally:
            _lib.set_tunable('upsum_bwd_band', 1)
    dy64 = dy.double().reshape(B, H, W, E).permute(0, 3, 1, 2)
    for k, f in enumerate((2, 4, 8)):
        z = torch.zeros(B, E, H // f, W // f, dtype=torch.float64, device=dev, requires_grad=True)
        F.interpolate(z, size=(H, W), mode='bilinear', align_corners=False).backward(dy64)
        ref = z.grad.permute(0, 2, 3, 1).reshape(B, -1, E)
        for mode in (1, 0):
            err = float((outs[mode][k].double() - ref).abs().max() / ref.abs().max())
            assert err < 2e-6, (f, mode, err)

"""The frozen SegFormer head's sum + BatchNorm(eval) + ReLU + linear_pred as one kernel (csrc/head_tail.hip; reference segformer_head.py:75-98)
against the fp64 form of the same operations, against the two-kernel route it replaces, and through SegFormerHead.forward (tap on linear_pred)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ref64(zs, sizes, fb, scale, shift, wp, bp):
    B, _, E = zs[0].shape
    H, W = sizes[0]
    tot = None
    for z, (h, w) in zip(zs, sizes):
        m = z.double().reshape(B, h, w, E).permute(0, 3, 1, 2)
        if (h, w) != (H, W):
            m = F.interpolate(m, size=(H, W), mode='bilinear', align_corners=False)
        tot = m if tot is None else tot + m
    if fb is not None:
        tot = tot + fb.double().view(1, -1, 1, 1)
    y = torch.relu(tot * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    return F.conv2d(y, wp.double().view(wp.shape[0], E, 1, 1), None if bp is None else bp.double())


@pytest.mark.parametrize('B,H,W,E,classes,bias', [(2, 128, 128, 768, 150, False), (1, 64, 96, 256, 19, True), (3, 8, 32, 64, 150, True),
                                                   (1, 16, 64, 128, 160, False), (2, 32, 32, 96, 1, True)])
def test_head_tail_matches_fp64_and_the_two_kernel_route(B, H, W, E, classes, bias):
    from segdistill_amd import headfuse
    from segdistill_amd.linear import linear_to_planes
    torch.manual_seed(H + E)
    sizes = [(H, W), (H // 2, W // 2), (H // 4, W // 4), (H // 8, W // 8)]
    zs = [torch.randn(B, h * w, E, device=DEV) for (h, w) in sizes]
    fb = torch.randn(E, device=DEV) if bias else None
    scale = torch.rand(E, device=DEV) + 0.5
    shift = torch.randn(E, device=DEV) * 0.5
    wp = torch.randn(classes, E, device=DEV) / E ** 0.5
    bp = torch.randn(classes, device=DEV) if bias else None
    with torch.no_grad():
        assert headfuse.head_tail_supported(zs, sizes, classes)
        out = headfuse.head_tail(zs, sizes, fb, scale, shift, wp, bp)
        ref = _ref64(zs, sizes, fb, scale, shift, wp, bp)
        y = headfuse.upsum_affine_inference(zs, fb, sizes, scale, shift, relu=True)
        two = F.linear(y.double(), wp.double(), None if bp is None else bp.double()).transpose(1, 2).reshape(B, classes, H, W)
    assert out.shape == ref.shape
    s = float(ref.abs().max())
    e_one = float((out.double() - ref).abs().max()) / s
    e_two = float((two - ref).abs().max()) / s          # the summed map in fp32, the product in fp64: the error the sum stage alone carries
    assert e_one < 3e-6, (e_one, e_two)


def test_segformer_head_takes_the_fused_tail_when_frozen(monkeypatch):
    from segdistill_amd import headfuse
    from segdistill_amd.decode_heads.segformer_head import SegFormerHead
    torch.manual_seed(1)
    head = SegFormerHead(feature_strides=[4, 8, 16, 32], in_channels=[64, 128, 320, 512], in_index=[0, 1, 2, 3], channels=128, dropout_ratio=0.1,
                         num_classes=150, norm_cfg=dict(type='BN', requires_grad=True), align_corners=False,
                         decoder_params=dict(embed_dim=256)).to(DEV).eval()
    with torch.no_grad():
        head.linear_fuse.norm.running_mean.normal_()
        head.linear_fuse.norm.running_var.uniform_(0.5, 2.0)
    for p in head.parameters():
        p.requires_grad_(False)
    feats = [torch.randn(2, c, 128 // f, 128 // f, device=DEV).contiguous(memory_format=torch.channels_last)
             for c, f in ((64, 1), (128, 2), (320, 4), (512, 8))]
    calls, seen = [], []
    real = headfuse.head_tail
    monkeypatch.setattr(headfuse, 'head_tail', lambda *a: (calls.append(1), real(*a))[1])
    head.linear_pred.register_forward_hook(lambda m, i, o: seen.append(o))
    with torch.no_grad():
        a = head(feats)
        assert calls == [1] and len(seen) == 1 and seen[0] is a
        monkeypatch.setattr(headfuse, '_HEAD_TAIL', False)
        b = head(feats)
        assert calls == [1] and len(seen) == 2
        monkeypatch.setattr(headfuse, '_HEAD_TAIL', True)
        h = head.linear_fuse.register_forward_hook(lambda m, i, o: None)       # a tap on the fused map: it has to exist
        head(feats)
        assert calls == [1]
        h.remove()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            head(feats)
        assert calls == [1]
    assert a.shape == b.shape == (2, 150, 128, 128)
    assert float((a - b).abs().max()) < 3e-6 * float(b.abs().max()) + 1e-6


def test_unsupported_shapes_are_refused():
    from segdistill_amd import _lib
    L = _lib.lib()
    assert not L.sd_head_tail_supported(128, 48, 768, 150)
    assert not L.sd_head_tail_supported(12, 32, 768, 150)
    assert not L.sd_head_tail_supported(128, 128, 760, 150)
    assert not L.sd_head_tail_supported(128, 128, 768, 161)
    z = torch.zeros(4096, device=DEV)
    p = z.data_ptr()
    assert L.sd_head_tail_f32(p, p, p, p, None, p, p, p, None, p, 1, 12, 32, 64, 8, None) == -6

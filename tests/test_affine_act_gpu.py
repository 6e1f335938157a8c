"""Eval-mode BatchNorm (+ residual) (+ ReLU) of a frozen convolutional network in one in-place pass (csrc/affine_act.hip; reference
backbones/resnet.py:18-100, ConvModule of psp_head.py:38-44): the C entry point against fp64, and frozen ResNetV1c / PSPHead forwards with and without it."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('shape', [(2, 64, 32, 32), (1, 37, 9, 7), (3, 256, 16, 16), (2, 8, 130, 66)])
@pytest.mark.parametrize('res', [False, True])
@pytest.mark.parametrize('relu', [False, True])
def test_entry_point_matches_fp64(shape, res, relu):
    from segdistill_amd import _lib
    L = _lib.lib()
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C)
    x = torch.randn(shape, generator=g).to(DEV)
    r = torch.randn(shape, generator=g).to(DEV) if res else None
    scale, shift = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
    ref = x.double() * scale.double().view(1, C, 1, 1) + shift.double().view(1, C, 1, 1)
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    y = x.clone()
    assert L.sd_affine_act_nchw(y.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), B * C, C, H * W,
                                int(relu), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert float((y.double() - ref).abs().max()) <= 1e-6 * (float(ref.abs().max()) + 1.0)
    assert L.sd_affine_act_nchw(y.data_ptr(), None, y.data_ptr(), scale.data_ptr(), shift.data_ptr(), B * C + 1, C, H * W, 0, None) == -2


def _randomise_norms(net, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.num_features, generator=g))


@pytest.mark.parametrize('depth', [18, 50])
def test_frozen_resnet_and_psp_head_match_the_three_launch_form(depth, monkeypatch):
    import segdistill_amd
    from segdistill_amd import affine_act
    from segdistill_amd.builder import BACKBONES, HEADS, build_from_cfg
    segdistill_amd.register_all()
    torch.manual_seed(depth)
    norm = dict(type='BN', requires_grad=True)
    net = build_from_cfg(dict(type='ResNetV1c', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 2, 4), strides=(1, 2, 1, 1),
                              norm_cfg=norm, norm_eval=False, style='pytorch', contract_dilation=True), BACKBONES)
    cin = 512 if depth == 18 else 2048
    head = build_from_cfg(dict(type='PSPHead', in_channels=cin, in_index=3, channels=64, pool_scales=(1, 2, 3, 6), dropout_ratio=0.1, num_classes=19,
                               norm_cfg=norm, align_corners=False, loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)), HEADS)
    _randomise_norms(net, 1)
    _randomise_norms(head, 2)
    net, head = net.to(DEV).eval(), head.to(DEV).eval()
    for p in list(net.parameters()) + list(head.parameters()):
        p.requires_grad_(False)
    x = torch.randn(2, 3, 96, 96, device=DEV)
    outs = {}
    calls = []
    real = affine_act.eval_norm_act_
    monkeypatch.setattr(affine_act, 'eval_norm_act_', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for flag in (True, False):
        monkeypatch.setattr(affine_act, '_ENABLED', flag)
        calls.clear()
        with torch.no_grad():
            feats = net(x)
            outs[flag] = [f.clone() for f in feats] + [head(feats).clone()]
        torch.cuda.synchronize()
        assert bool(calls) == flag
    for a, b in zip(outs[True], outs[False]):
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-6)
    # a network that is being trained never takes the in-place path
    monkeypatch.setattr(affine_act, '_ENABLED', True)
    calls.clear()
    net.train()
    net(x)
    assert not calls

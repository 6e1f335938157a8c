"""SegFormerHead while TRAINING (nobody taps linear_c1..4): each branch's Linear and its block of linear_fuse are ONE token GEMM on their product
(decode_heads/segformer_head.py::_fused_sum, round 6; reference segformer_head.py:75-98 runs the four MLPs, three resizes, a concat and the 1x1
fuse conv) -- same logits and same parameter gradients as the two-GEMM form, which the reference-generated train-step fixtures pin."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_folded_branches_equal_the_two_gemm_form(monkeypatch):
    import segdistill_amd
    from segdistill_amd import deferred
    from segdistill_amd.builder import HEADS, build_from_cfg
    from segdistill_amd.decode_heads import segformer_head
    segdistill_amd.register_all()
    torch.manual_seed(0)
    head = build_from_cfg(dict(type='SegFormerHead', in_channels=[32, 64, 160, 256], in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32], channels=128,
                               dropout_ratio=0.0, num_classes=150, norm_cfg=dict(type='SyncBN', requires_grad=True), align_corners=False,
                               decoder_params=dict(embed_dim=256), loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
                          HEADS).to(DEV).train()
    B = 4
    feats0 = [torch.randn(B, (128 >> i) ** 2, c, device=DEV) for i, c in enumerate((32, 64, 160, 256))]
    g = torch.randn(B, 150, 128, 128, device=DEV)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(segformer_head, '_FOLD_TRAIN', flag)
        for p in head.parameters():
            p.grad = None
        leaves = [f.clone().requires_grad_(True) for f in feats0]
        feats = [f.reshape(B, 128 >> i, 128 >> i, f.shape[-1]).permute(0, 3, 1, 2) for i, f in enumerate(leaves)]
        y = head(feats)
        with deferred.scope():
            y.backward(g)
        torch.cuda.synchronize()
        out[flag] = (y.detach().clone(), [f.grad.clone() for f in leaves], {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None})

    def rel(a, b):
        # rel-L2: a ReLU whose input sits within rounding of zero may open on one side only -- single elements then differ by O(1) while the
        # tensors agree (against fp64 both forms show the same 1-6 % max-norm outliers and 1e-4 rel-L2)
        a, b = a.double(), b.double()
        return float(((a - b).pow(2).sum() / (b.pow(2).sum() + 1e-30)).sqrt())
    assert float((out[True][0].double() - out[False][0].double()).abs().max() / out[False][0].double().abs().max()) < 2e-5
    for a, b in zip(out[True][1], out[False][1]):
        assert rel(a, b) < 2e-3
    assert out[True][2].keys() == out[False][2].keys()
    for n in out[True][2]:
        if n.endswith('.proj.bias'):          # mathematically zero (a constant in front of the BatchNorm): rounding noise on both sides
            continue
        assert rel(out[True][2][n], out[False][2][n]) < 2e-3, n

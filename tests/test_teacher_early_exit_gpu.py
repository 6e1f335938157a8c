"""The frozen teacher's forward stops at its last tap (segmentors/sd_module.py::_teacher_forward, distillation/opts.py::TapsComplete; round 6).
The reference runs the teacher's whole forward_train and discards what nobody taps (SD_structure.py:70-75): the KD losses must not change when
the layers behind the last tap are skipped, and a tap on the last module skips nothing."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _model(taps):
    import warnings
    import segdistill_amd
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.segmentors import sd_module
    segdistill_amd.register_all()
    norm = dict(type='SyncBN', requires_grad=True)
    seg = dict(type='EncoderDecoder', pretrained=None, backbone=dict(type='mit_b0', style='pytorch'),
               decode_head=dict(type='SegFormerHead', in_channels=[32, 64, 160, 256], in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32],
                                channels=128, dropout_ratio=0.1, num_classes=150, norm_cfg=norm, align_corners=False,
                                decoder_params=dict(embed_dim=256), loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    cfg = dict(type='SDModule', cfg_s=seg, cfg_t=seg, train_cfg=dict(), test_cfg=dict(mode='whole'), t_pretrain=None,
               distillation=[dict(student_layer=t, teacher_layer=t, loss_name='KLDLoss',
                                  loss_config=dict(alpha=1, tau=1, transform_config={'loss_type': 'channel', 'group_size': 8})) for t in taps])
    sd_module.SYNTHETIC_WEIGHTS_OK = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.manual_seed(0)
        m = build_segmentor(cfg).to(DEV)
    m.student.backbone.reset_drop_path(0.)
    return m.train()


@pytest.mark.parametrize('taps,skipped', [(['decode_head.linear_c4', 'decode_head.linear_c1'], True), (['decode_head.linear_pred'], False)])
def test_losses_do_not_change_and_the_tail_is_skipped_only_behind_the_last_tap(taps, skipped, monkeypatch):
    from segdistill_amd.segmentors import sd_module
    m = _model(taps)
    m.teacher_on_side_stream = False
    img = torch.randn(2, 3, 64, 64, device=DEV)
    gt = torch.randint(0, 150, (2, 1, 64, 64), device=DEV)
    calls = []
    m.teacher.decode_head.linear_pred.register_forward_pre_hook(lambda mod, i: calls.append(1))       # before the module runs (the tap's own hook raises behind it)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(sd_module, '_EARLY_EXIT', flag)
        calls.clear()
        torch.manual_seed(3)            # the student head's dropout
        losses = m.forward_train(img, None, gt)
        torch.cuda.synchronize()
        out[flag] = {k: float(v.mean()) for k, v in losses.items()}
        assert bool(calls) == (not (flag and skipped)), (flag, calls)
    assert out[True].keys() == out[False].keys()
    for k in out[True]:
        assert out[True][k] == pytest.approx(out[False][k], rel=1e-6, abs=1e-7), k

"""Generate tests/golden/kd_train_step.npz by RUNNING THE REFERENCE's SDModule.train_step
(imported from /root/reference through oracle/refimport.py) on closed-form weights and inputs.

Run in the build container only:   python -m oracle.gen_golden_nets
Recorded per case: the reference's log_vars of one train_step (n_iter = 1) and checksums of the
student gradients after loss.backward().  Determinism: student drop-path reset to 0, head
dropout_ratio 1e-12, teacher in eval mode (the evident intent -- SURVEY.md Q1), BN batch
statistics are deterministic.  TEST INFRASTRUCTURE; no reference source is written anywhere.
"""
from __future__ import annotations

import copy
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

from . import refimport
from .inputs import fill_state_dict_, wavy_image

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
NORM = dict(type='SyncBN', requires_grad=True)


def segformer(variant, ch, e):
    return dict(type='EncoderDecoder', pretrained=None, backbone=dict(type=f'mit_{variant}', style='pytorch'),
                decode_head=dict(type='SegFormerHead', in_channels=ch, in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32], channels=128,
                                 dropout_ratio=1e-12, num_classes=150, norm_cfg=NORM, align_corners=False, decoder_params=dict(embed_dim=e),
                                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))


def pspnet(depth, c4, c3, head_ch, aux_ch):
    return dict(type='EncoderDecoder', pretrained=None,
                backbone=dict(type='ResNetV1c', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 2, 4), strides=(1, 2, 1, 1),
                              norm_cfg=NORM, norm_eval=False, style='pytorch', contract_dilation=True),
                decode_head=dict(type='PSPHead', in_channels=c4, in_index=3, channels=head_ch, pool_scales=(1, 2, 3, 6), dropout_ratio=1e-12,
                                 num_classes=150, norm_cfg=NORM, align_corners=False,
                                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
                auxiliary_head=dict(type='FCNHead', in_channels=c3, in_index=2, channels=aux_ch, num_convs=1, concat_input=False,
                                    dropout_ratio=1e-12, num_classes=150, norm_cfg=NORM, align_corners=False,
                                    loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)))


CASES = {
    # name: (cfg_s, cfg_t, distillation, image shape)
    'segformer_b0_b0_cgd': (segformer('b0', [32, 64, 160, 256], 256), segformer('b0', [32, 64, 160, 256], 256),
                            [dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='CGDLoss',
                                  loss_config={'group_size': 8, 'alpha': 3, 'tau': 4})], (2, 3, 64, 64)),
    'pspnet_r18_r18_cd': (pspnet(18, 512, 256, 128, 64), pspnet(18, 512, 256, 128, 64),
                          [dict(student_layer='decode_head.conv_seg', teacher_layer='decode_head.conv_seg', loss_name='CDLoss', loss_config={})],
                          (2, 3, 64, 64)),
}

PROBE_PARAMS = {
    'segformer_b0_b0_cgd': ['student.decode_head.linear_pred.weight', 'student.decode_head.linear_fuse.conv.weight',
                            'student.backbone.block4.1.mlp.fc2.weight', 'student.backbone.patch_embed1.proj.weight',
                            'student.backbone.block1.0.attn.sr.weight', 'student.backbone.block2.0.mlp.dwconv.dwconv.weight'],
    'pspnet_r18_r18_cd': ['student.decode_head.conv_seg.weight', 'student.decode_head.bottleneck.conv.weight',
                          'student.backbone.layer4.1.conv2.weight', 'student.backbone.stem.0.weight', 'student.auxiliary_head.conv_seg.weight'],
}


def build_reference_case(ns, name):
    cfg_s, cfg_t, distill, shape = CASES[name]
    # the reference insists on torch.load(t_pretrain) (SD_structure.py:36-37): give it a real (empty) checkpoint file
    with tempfile.NamedTemporaryFile(suffix='.pth', delete=False) as f:
        torch.save({'state_dict': {}}, f.name)
        ck = f.name
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ns.SDModule(cfg_s=copy.deepcopy(cfg_s), cfg_t=copy.deepcopy(cfg_t), train_cfg=None, test_cfg=dict(mode='whole'),
                            distillation=copy.deepcopy(distill), t_pretrain=ck)
    os.unlink(ck)
    fill_state_dict_(model.student, salt=0)
    fill_state_dict_(model.teacher, salt=1)
    if hasattr(model.student.backbone, 'reset_drop_path'):
        model.student.backbone.reset_drop_path(0.)
    model.train()
    model.teacher.eval()
    return model, shape


def run_case(ns, name, dtype):
    model, shape = build_reference_case(ns, name)
    model = model.to(dtype)
    img, lab = wavy_image(shape)
    batch = dict(img=torch.tensor(img).to(dtype), img_metas=None, gt_semantic_seg=torch.tensor(lab))
    res = model.train_step(batch, None)
    res['loss'].backward()
    named = dict(model.named_parameters())
    vals = {f'log/{k}': float(v) for k, v in res['log_vars'].items()}
    for p in PROBE_PARAMS[name]:
        g = named[p].grad.double()
        vals[f'grad_abs_sum/{p}'] = g.abs().sum().item()
        vals[f'grad_l2/{p}'] = g.pow(2).sum().sqrt().item()
    vals['grad_l2_total'] = torch.sqrt(sum(p.grad.double().pow(2).sum() for n, p in named.items()
                                           if p.grad is not None and n.startswith('student.'))).item()
    return vals, shape


def main():
    """The reference is run twice per case: in fp64 (stored as THE expected values) and in fp32 (only its
    relative deviation from fp64 is stored, key suffix '@fp32dev', so that fp32 consumers can set
    tolerances from the reference's own rounding instead of guessing)."""
    ns = refimport.load_full()
    out = {}
    for name in CASES:
        v64, shape = run_case(ns, name, torch.float64)
        v32, _ = run_case(ns, name, torch.float32)
        for k, v in v64.items():
            out[f'{name}/{k}'] = np.float64(v)
            out[f'{name}/{k}@fp32dev'] = np.float64(abs(v32[k] - v) / max(abs(v), 1e-300))
        out[f'{name}/shape'] = np.asarray(shape)
        print(name, {k: round(v, 6) for k, v in v64.items() if k.startswith('log/')})
        print('   fp32 deviation of the reference from its own fp64:', {k.split('/')[-1][-28:]: f'{abs(v32[k] - v) / max(abs(v), 1e-300):.1e}' for k, v in v64.items()})
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'kd_train_step.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    sys.exit(main())

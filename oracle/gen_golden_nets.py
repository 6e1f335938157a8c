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
from .inputs import fill_state_dict_, fill_state_dict_hashed_, probe_vector, wavy_image

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


def swin_upernet(embed, depths, heads, head_ch, aux_ch, ape=False, pretrain=224):
    """A narrow Swin + UPerHead (+ FCN auxiliary head): the layout of the reference's local_configs/_base_/models/upernet_swin.py."""
    dims = [embed * 2 ** i for i in range(4)]
    return dict(type='EncoderDecoder', pretrained=None,
                backbone=dict(type='SwinTransformer', pretrain_img_size=pretrain, embed_dim=embed, depths=depths, num_heads=heads, window_size=7,
                              mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.3, ape=ape,
                              patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False),
                decode_head=dict(type='UPerHead', in_channels=dims, in_index=[0, 1, 2, 3], pool_scales=(1, 2, 3, 6), channels=head_ch,
                                 dropout_ratio=0.1, num_classes=150, norm_cfg=NORM, align_corners=False,
                                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
                auxiliary_head=dict(type='FCNHead', in_channels=dims[2], in_index=2, channels=aux_ch, num_convs=1, concat_input=False,
                                    dropout_ratio=0.1, num_classes=150, norm_cfg=NORM, align_corners=False,
                                    loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)))


_CGD84 = {'group_size': 8, 'alpha': 3, 'tau': 4}
_SWIN_TAPS = ['backbone', 'decode_head.fpn_bottleneck', 'decode_head.bottleneck', 'decode_head.conv_seg']
CASES = {
    # name: (cfg_s, cfg_t, distillation, image shape)
    'segformer_b0_b0_cgd': (segformer('b0', [32, 64, 160, 256], 256), segformer('b0', [32, 64, 160, 256], 256),
                            [dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='CGDLoss',
                                  loss_config=_CGD84)], (2, 3, 64, 64)),
    # hash-uniform weights (OPTIONS below): well conditioned, so the fp32 GPU run can be held to the 1e-3 bar
    'pspnet_r18_r18_cd': (pspnet(18, 512, 256, 128, 64), pspnet(18, 512, 256, 128, 64),
                          [dict(student_layer='decode_head.conv_seg', teacher_layer='decode_head.conv_seg', loss_name='CDLoss', loss_config={})],
                          (2, 3, 64, 64)),
    # Swin + UPerHead teachers (the cfg4 teacher family, reference swin_transformer.py:440-618, uper_head.py:12-126).  96x64 -> 24x16
    # tokens: every stage pads its map to a multiple of the 7x7 window (28x21, 14x14, 7x7, 7x7) and odd blocks shift it
    'segformer_b0_swin_uper_cgd': (segformer('b0', [32, 64, 160, 256], 256), swin_upernet(24, [2, 2, 2, 2], [3, 6, 12, 24], 32, 16),
                                   [dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.conv_seg', loss_name='CGDLoss',
                                         loss_config=_CGD84)], (2, 3, 96, 64)),
    # 112x112 -> 28x28 tokens: window-aligned at stages 1-2 (no padding, shift only); absolute position embedding (bicubic 16x16 -> 28x28)
    # and PatchMerging on the odd 7x7 map of stage 3; a PSPNet student as in cfg4
    'pspnet_r18_swin_ape_uper_cd': (pspnet(18, 512, 256, 128, 64), swin_upernet(16, [2, 2, 4, 2], [2, 4, 8, 16], 48, 16, ape=True, pretrain=64),
                                    [dict(student_layer='decode_head.conv_seg', teacher_layer='decode_head.conv_seg', loss_name='CDLoss',
                                          loss_config={})], (2, 3, 112, 112)),
}

# per case: which weight filler (oracle/inputs.py) and which TEACHER taps are summarised into the fixture
OPTIONS = {
    'segformer_b0_b0_cgd': dict(filler='sine', taps=[]),
    'pspnet_r18_r18_cd': dict(filler='hashed', taps=['decode_head.conv_seg']),
    'segformer_b0_swin_uper_cgd': dict(filler='hashed', taps=_SWIN_TAPS),
    'pspnet_r18_swin_ape_uper_cd': dict(filler='hashed', taps=_SWIN_TAPS),
}

_SEGF_PROBES = ['student.decode_head.linear_pred.weight', 'student.decode_head.linear_fuse.conv.weight',
                'student.backbone.block4.1.mlp.fc2.weight', 'student.backbone.patch_embed1.proj.weight']
_PSP_PROBES = ['student.decode_head.conv_seg.weight', 'student.decode_head.bottleneck.conv.weight',
               'student.backbone.layer4.1.conv2.weight', 'student.backbone.stem.0.weight', 'student.auxiliary_head.conv_seg.weight']
PROBE_PARAMS = {
    'segformer_b0_swin_uper_cgd': _SEGF_PROBES,
    'pspnet_r18_swin_ape_uper_cd': _PSP_PROBES,
    'segformer_b0_b0_cgd': ['student.decode_head.linear_pred.weight', 'student.decode_head.linear_fuse.conv.weight',
                            'student.backbone.block4.1.mlp.fc2.weight', 'student.backbone.patch_embed1.proj.weight',
                            'student.backbone.block1.0.attn.sr.weight', 'student.backbone.block2.0.mlp.dwconv.dwconv.weight'],
    'pspnet_r18_r18_cd': ['student.decode_head.conv_seg.weight', 'student.decode_head.bottleneck.conv.weight',
                          'student.backbone.layer4.1.conv2.weight', 'student.backbone.stem.0.weight', 'student.auxiliary_head.conv_seg.weight'],
}


def filler_of(name):
    return fill_state_dict_hashed_ if OPTIONS[name]['filler'] == 'hashed' else fill_state_dict_


def tap_summary(t):
    """What the fixture keeps of a tapped tensor: its L2 norm, its mean, its inner product with a fixed probe direction and a small corner."""
    a = t.detach().double().cpu().numpy()
    corner = a[0, :4, :3, :3] if a.ndim == 4 else a.reshape(-1)[:36]
    return {'l2': float(np.sqrt((a ** 2).sum())), 'mean': float(a.mean()), 'probe': float((a * probe_vector(a.shape)).sum()),
            'corner': np.ascontiguousarray(corner), 'shape': np.asarray(a.shape)}


def record_taps(net, names, sink):
    """Forward hooks on `net`'s modules `names`; tuple outputs (a backbone's feature pyramid) are recorded per level as name[i]."""
    mods = dict(net.named_modules())
    handles = []
    for n in names:
        def hook(_m, _inp, out, n=n):
            if isinstance(out, (tuple, list)):
                for i, o in enumerate(out):
                    sink[f'{n}[{i}]'] = tap_summary(o)
            else:
                sink[n] = tap_summary(out)
        handles.append(mods[n].register_forward_hook(hook))
    return handles


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
    fill = filler_of(name)
    fill(model.student, salt=0)
    fill(model.teacher, salt=1)
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
    sink = {}
    handles = record_taps(model.teacher, OPTIONS[name]['taps'], sink)
    res = model.train_step(batch, None)
    for h in handles:
        h.remove()
    res['loss'].backward()
    named = dict(model.named_parameters())
    vals = {f'log/{k}': float(v) for k, v in res['log_vars'].items()}
    for tname, summ in sink.items():
        for k, v in summ.items():
            vals[f'tap/{tname}/{k}'] = v
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
        dev = {}
        for k, v in v64.items():
            if isinstance(v, np.ndarray):
                out[f'{name}/{k}'] = v
                continue
            out[f'{name}/{k}'] = np.float64(v)
            dev[k] = abs(v32[k] - v) / max(abs(v), 1e-300)
            out[f'{name}/{k}@fp32dev'] = np.float64(dev[k])
        out[f'{name}/shape'] = np.asarray(shape)
        print(name, {k: round(v, 6) for k, v in v64.items() if k.startswith('log/')})
        print('   fp32 deviation of the reference from its own fp64:', {k[-34:]: f'{d:.1e}' for k, d in dev.items()})
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'kd_train_step.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    sys.exit(main())

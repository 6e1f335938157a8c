"""tools/train.py on the GPU with a directory dataset feeding the full KD step (HIP criteria, hipGraph replay after warm-up)."""
import os
import subprocess
import sys

import pytest

from test_data_pipeline_cpu import ROOT, _fake_ade

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('graph', ['on', 'off'])
def test_kd_training_from_directory_dataset(tmp_path, graph):
    root = _fake_ade(str(tmp_path / 'ade'), n=6, size=(140, 180))
    cfg = tmp_path / 'kd_small.py'
    cfg.write_text(f'''
_base_ = ['{ROOT}/configs/kd/cfg2_segformer_b2_b0_cgd_ade20k.py']
crop_size = (128, 128)
train_pipeline = [
    dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', reduce_zero_label=True),
    dict(type='Resize', img_scale=(512, 128), ratio_range=(0.5, 2.0)), dict(type='RandomCrop', crop_size=crop_size, cat_max_ratio=0.75),
    dict(type='RandomFlip', prob=0.5), dict(type='PhotoMetricDistortion'),
    dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
    dict(type='Pad', size=crop_size, pad_val=0, seg_pad_val=255), dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_semantic_seg'])]
data = dict(_delete_=True, samples_per_gpu=2, workers_per_gpu=2,
            train=dict(type='RepeatDataset', times=4, dataset=dict(type='ADE20KDataset', data_root='unused', img_dir='images/training',
                                                                   ann_dir='annotations/training', pipeline=train_pipeline)))
log_config = dict(interval=2)
''')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'train.py'), str(cfg), '--data-root', root, '--iters', '8', '--work-dir',
                        str(tmp_path / 'work'), '--synthetic-weights', '--graph', graph, '--options', 'model.cfg_t.backbone.type=mit_b0',
                        'model.cfg_t.decode_head.in_channels=[32,64,160,256]', 'model.cfg_t.decode_head.decoder_params.embed_dim=256'],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('Iter [')]
    assert len(lines) == 4, r.stdout[-1500:]
    last = dict(kv.split(': ') for kv in lines[-1].split('\t')[1].split(', '))
    assert any(k.startswith('loss') and k != 'loss' for k in last), last          # the KD term is logged next to decode.loss_seg
    assert all(float(v) == float(v) for v in last.values())          # no NaN

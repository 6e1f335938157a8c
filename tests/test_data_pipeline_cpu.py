"""The ADE20K training pipeline (segdistill_amd/data): transform semantics of the reference's pipeline classes
(mmseg/datasets/pipelines/transforms.py, loading.py, formating.py), dataset scanning, per-rank sharding, and an end-to-end
run of tools/train.py on a tiny ADE-shaped directory tree.  cv2 is not installed, so image arithmetic is checked against
hand-computed values and OpenCV's documented conventions, not against cv2 itself."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_ade(root, n=6, size=(40, 56), seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    for sub in ('images/training', 'annotations/training'):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in range(n):
        h, w = size[0] + 3 * i, size[1] + 2 * i
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        seg = rng.randint(0, 151, (h, w)).astype(np.uint8)          # 0 = "other", 1..150 = classes
        seg[: h // 4] = 7                                           # a dominant region for the cat_max_ratio logic
        Image.fromarray(img).save(os.path.join(root, 'images/training', f'ADE_train_{i:08d}.jpg'), quality=95)
        Image.fromarray(seg).save(os.path.join(root, 'annotations/training', f'ADE_train_{i:08d}.png'))
    return root


def test_imresize_follows_opencv_sampling():
    from segdistill_amd.data import imops
    img = np.arange(4 * 6, dtype=np.uint8).reshape(4, 6) * 10
    up = imops.imresize(img, (12, 8), 'bilinear')                    # x2: src = (dst + 0.5)/2 - 0.5, edge clamp
    assert up.shape == (8, 12)
    assert up[0, 0] == img[0, 0] and up[-1, -1] == img[-1, -1]       # clamped corners
    # 8-bit images go through OpenCV's fixed-point arithmetic (11-bit coefficients, two truncating shifts, + 2 >> 2): lambda = .25 at dst x = 1
    # gives 0.75 * 0 + 0.25 * 10 = 2.5 -> ((2048 * ((10 * 512) >> 4)) >> 16) + 2 >> 2 = (10 + 2) >> 2 = 3, and 7.5 -> (30 + 2) >> 2 = 8
    assert up[0, 1] == 3 and up[0, 2] == 8
    # hand-computed known answer, 4 x 4 -> 7 wide x 5 high.  Element [1][1]: fx = 1.5 * 4/7 - .5 = .35714 -> a = rint(.64286 * 2048) = 1317,
    # b = rint(.35714 * 2048) = 731; fy = 1.5 * .8 - .5 = .7 -> c = rint(.3 * 2048) = 614, d = rint(.7 * 2048) = 1434;
    # R0 = (3 * 1317 + 19 * 731) >> 4 = 17840 >> 4 = 1115, R1 = (67 * 1317 + 83 * 731) >> 4 = 148912 >> 4 = 9307;
    # ((614 * 1115) >> 16) + ((1434 * 9307) >> 16) = 10 + 203 = 213; (213 + 2) >> 2 = 53   (the exact value is 53.51: the shifts truncate)
    kat = np.arange(16, dtype=np.uint8).reshape(4, 4) * 16 + 3
    want = [[3, 9, 18, 27, 36, 45, 51], [48, 53, 63, 72, 81, 90, 96], [99, 105, 114, 123, 132, 141, 147], [150, 156, 165, 174, 183, 192, 198],
            [195, 201, 210, 219, 228, 237, 243]]
    got = imops.imresize(kat, (7, 5), 'bilinear')
    assert got.tolist() == want
    assert np.array_equal(imops.imresize(np.stack([kat] * 3, -1), (7, 5))[..., 1], got)          # channels are independent
    # the crop-aware form computes a window of the same resize: bit-identical to resize-then-slice
    rng = np.random.default_rng(0)
    big = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    full = imops.imresize(big, (91, 70))
    lazy = imops.LazyResize(big, (91, 70))
    assert lazy.shape == full.shape and np.array_equal(np.asarray(lazy), full)
    assert np.array_equal(lazy.region(11, 43, 20, 91), full[11:43, 20:91]) and np.array_equal(lazy.region(60, 200, -5, 30), full[60:, :30])
    near = imops.imresize(img, (3, 2), 'nearest')                    # floor(dst * scale): rows 0,2  cols 0,2,4
    assert np.array_equal(near, img[[0, 2]][:, [0, 2, 4]])
    assert imops.rescale_size((683, 512), (2048, 512)) == (683, 512)           # short edge already at the bound
    assert imops.rescale_size((640, 480), (1024, 256)) == (341, 256)           # min(1024/640, 256/480)
    assert imops.imrescale(np.zeros((480, 640, 3), np.uint8), (1024, 256)).shape == (256, 341, 3)
    lab = np.array([[0, 255], [3, 150]], np.uint8)
    assert set(np.unique(imops.imresize(lab, (5, 7), 'nearest'))) <= {0, 3, 150, 255}      # labels are never blended


def test_hsv_conversion_conventions():
    from segdistill_amd.data import imops
    bgr = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [128, 128, 128], [0, 0, 0], [0, 255, 255]]], np.uint8)
    hsv = imops.bgr2hsv(bgr)[0]
    assert hsv[0].tolist() == [0, 255, 255] and hsv[1].tolist() == [60, 255, 255] and hsv[2].tolist() == [120, 255, 255]   # H in [0,180)
    assert hsv[3].tolist() == [0, 0, 128] and hsv[4].tolist() == [0, 0, 0] and hsv[5].tolist() == [30, 255, 255]
    assert np.array_equal(imops.hsv2bgr(imops.bgr2hsv(bgr)), bgr)
    rnd = np.random.RandomState(1).randint(0, 256, (32, 32, 3), dtype=np.uint8)
    assert np.abs(imops.hsv2bgr(imops.bgr2hsv(rnd)).astype(int) - rnd.astype(int)).max() <= 6    # 2-degree hue quantisation


def test_load_annotations_reduce_zero_label(tmp_path):
    from PIL import Image
    from segdistill_amd.data.pipelines import LoadAnnotations
    seg = np.array([[0, 1, 2], [150, 149, 0]], np.uint8)
    Image.fromarray(seg).save(tmp_path / 'a.png')
    out = LoadAnnotations(reduce_zero_label=True)(dict(ann_info=dict(seg_map='a.png'), seg_prefix=str(tmp_path), seg_fields=[]))
    assert out['gt_semantic_seg'].tolist() == [[255, 0, 1], [149, 148, 255]] and out['seg_fields'] == ['gt_semantic_seg']
    out = LoadAnnotations()(dict(ann_info=dict(seg_map='a.png'), seg_prefix=str(tmp_path), seg_fields=[]))
    assert np.array_equal(out['gt_semantic_seg'], seg)


def test_geometric_transforms_keep_image_and_labels_aligned():
    from segdistill_amd.data.pipelines import Pad, RandomCrop, RandomFlip, Resize
    h, w = 30, 50
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([yy, xx, yy + xx], -1).astype(np.uint8)
    seg = (xx % 7).astype(np.uint8)
    np.random.seed(3)
    r = dict(img=img, gt_semantic_seg=seg, seg_fields=['gt_semantic_seg'])
    r = Resize(img_scale=(100, 60), ratio_range=(1.0, 1.0))(r)                 # keep_ratio rescale by min(100/50, 60/30) = 2
    assert r['img'].shape[:2] == (60, 100) == r['gt_semantic_seg'].shape and r['scale'] == (100, 60)
    assert np.allclose(r['scale_factor'], 2.0)
    r = RandomCrop((32, 32), cat_max_ratio=0.75)(r)
    assert r['img'].shape[:2] == (32, 32) == r['gt_semantic_seg'].shape
    x0 = int(r['img'][0, 0, 1])                                                 # channel 1 encodes the source column (x2 upsample)
    assert abs(int(r['gt_semantic_seg'][0, 0]) - ((x0 // 1) % 7)) <= 6         # still label values
    before_img, before_seg = r['img'].copy(), r['gt_semantic_seg'].copy()
    r['flip'] = True
    r = RandomFlip(prob=0.5)(r)
    assert np.array_equal(r['img'], before_img[:, ::-1]) and np.array_equal(r['gt_semantic_seg'], before_seg[:, ::-1])
    r = Pad(size=(40, 48), pad_val=0, seg_pad_val=255)(r)
    assert r['img'].shape == (40, 48, 3) and (r['img'][32:] == 0).all() and (r['gt_semantic_seg'][:, 32:] == 255).all()
    assert np.array_equal(r['gt_semantic_seg'][:32, :32], before_seg[:, ::-1])
    with pytest.raises(ValueError):
        Pad(size=(8, 8))(dict(img=img, seg_fields=[]))


def test_random_crop_respects_cat_max_ratio():
    from segdistill_amd.data.pipelines import RandomCrop
    seg = np.zeros((64, 64), np.uint8)
    seg[:, 48:] = np.arange(16, dtype=np.uint8)[None, :] + 1          # only the right quarter is diverse
    img = np.zeros((64, 64, 3), np.uint8)
    np.random.seed(0)
    hits = 0
    for _ in range(40):
        out = RandomCrop((16, 16), cat_max_ratio=0.75)(dict(img=img, gt_semantic_seg=seg, seg_fields=['gt_semantic_seg']))
        labels, cnt = np.unique(out['gt_semantic_seg'], return_counts=True)
        hits += len(cnt) > 1 and cnt.max() / cnt.sum() < 0.75
    assert hits >= 30            # a single draw lands on the diverse quarter ~25 % of the time; 10 re-draws raise that to ~95 %


def test_photometric_distortion_and_normalize():
    from segdistill_amd.data.pipelines import Normalize, PhotoMetricDistortion
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (24, 24, 3), dtype=np.uint8)
    np.random.seed(5)
    changed = 0
    for _ in range(20):
        out = PhotoMetricDistortion()(dict(img=img.copy()))['img']
        assert out.dtype == np.uint8 and out.shape == img.shape
        changed += not np.array_equal(out, img)
    assert 12 <= changed <= 20                                         # identity only when all four coins say "skip" (1/16)
    pm = PhotoMetricDistortion()
    assert pm.convert(np.array([250, 10], np.uint8), alpha=1.5, beta=-20).tolist() == [255, 0]
    n = Normalize(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)(dict(img=img))['img']
    assert n.dtype == np.float32
    assert np.allclose(n[..., 0], (img[..., 2].astype(np.float32) - 123.675) / 58.395, atol=1e-5)      # BGR -> RGB first


def test_dataset_loader_and_sharding(tmp_path):
    from segdistill_amd.config import Config
    from segdistill_amd.data import build_dataloader, build_dataset
    root = _fake_ade(str(tmp_path / 'ade'))
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', '_base_', 'ade20k.py'))
    import copy
    tcfg = copy.deepcopy(dict(cfg.data.train))
    tcfg['dataset']['data_root'] = root
    for t in tcfg['dataset']['pipeline']:
        if t['type'] == 'Resize':
            t['img_scale'] = (128, 32)
        if t['type'] in ('RandomCrop',):
            t['crop_size'] = (32, 32)
        if t['type'] == 'Pad':
            t['size'] = (32, 32)
    tcfg['times'] = 3
    ds = build_dataset(tcfg)
    assert len(ds) == 18 and len(ds.dataset) == 6
    assert ds.dataset.img_infos[0] == dict(filename='ADE_train_00000000.jpg', ann=dict(seg_map='ADE_train_00000000.png'))
    np.random.seed(0)
    item = ds[7]
    assert item['img'].shape == (3, 32, 32) and item['img'].dtype == torch.float32
    assert item['gt_semantic_seg'].shape == (1, 32, 32) and item['gt_semantic_seg'].dtype == torch.int64
    labels = set(item['gt_semantic_seg'].unique().tolist())
    assert labels <= set(range(150)) | {255}
    loader = build_dataloader(ds, 4, 0, seed=1)
    batch = next(iter(loader))
    assert batch['img'].shape == (4, 3, 32, 32) and batch['gt_semantic_seg'].shape == (4, 1, 32, 32) and len(batch['img_metas']) == 4
    # two ranks see disjoint halves of every epoch
    seen = []
    for rank in range(2):
        ld = build_dataloader(ds, 3, 0, world=2, rank=rank, seed=1)
        seen.append(list(iter(ld.sampler)))
    assert len(seen[0]) == len(seen[1]) == 9 and not (set(seen[0]) & set(seen[1])) and sorted(seen[0] + seen[1]) == list(range(18))
    with pytest.raises(FileNotFoundError):
        bad = dict(tcfg['dataset'], data_root=str(tmp_path / 'missing'))
        build_dataset(bad)


def test_train_tool_runs_on_a_directory_dataset(tmp_path):
    """tools/train.py end to end on CPU: the SDModule config with data.train pointing at a fake ADE tree.  The distillation list
    is emptied for this run -- the KD criteria are HIP-only by design (no CPU path) -- so it exercises dataset -> loader ->
    SDModule.train_step (student CE) -> optimizer -> checkpoint."""
    root = _fake_ade(str(tmp_path / 'ade'), n=4, size=(70, 90))
    cfg = tmp_path / 'kd_tiny.py'
    cfg.write_text(f'''
_base_ = ['{ROOT}/configs/kd/cfg2_segformer_b2_b0_cgd_ade20k.py']
crop_size = (64, 64)
train_pipeline = [
    dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', reduce_zero_label=True),
    dict(type='Resize', img_scale=(256, 64), ratio_range=(0.5, 2.0)), dict(type='RandomCrop', crop_size=crop_size, cat_max_ratio=0.75),
    dict(type='RandomFlip', prob=0.5), dict(type='PhotoMetricDistortion'),
    dict(type='Normalize', mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
    dict(type='Pad', size=crop_size, pad_val=0, seg_pad_val=255), dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_semantic_seg'])]
data = dict(_delete_=True, samples_per_gpu=2, workers_per_gpu=0,
            train=dict(type='RepeatDataset', times=2, dataset=dict(type='ADE20KDataset', data_root='unused', img_dir='images/training',
                                                                   ann_dir='annotations/training', pipeline=train_pipeline)))
log_config = dict(interval=1)
''')
    env = dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'train.py'), str(cfg), '--data-root', root, '--iters', '3', '--work-dir',
                        str(tmp_path / 'work'), '--synthetic-weights', '--options', 'model.cfg_t.backbone.type=mit_b0',
                        'model.cfg_t.decode_head.in_channels=[32,64,160,256]', 'model.cfg_t.decode_head.decoder_params.embed_dim=256', 'model.distillation=[]'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert 'dataset: 8 samples (RepeatDataset)' in r.stdout
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('Iter [')]
    assert len(lines) == 3 and 'loss' in lines[-1]
    assert os.path.isfile(tmp_path / 'work' / 'latest.pth')

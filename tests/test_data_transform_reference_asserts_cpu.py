"""The assertions of the reference's OWN pipeline tests (tests/test_data/test_transform.py: test_resize :13-103, test_flip :106-140,
test_random_crop :143-170, test_pad :173-207, test_normalize :250-275, test_seg_rescale :462-478), restated for segdistill_amd/data/pipelines.py.
The reference's fixture images tests/data/color.jpg (288 x 512 x 3) and seg.png are not part of the reference tree given here, so an image of
the same shape is synthesised; every expectation below is the reference's (constructor assertions, resulting shapes, round trips, the
normalisation formula) -- none depends on the pixel values of the missing files.  (The transforms of that file that are not on the ADE20K
training pipeline -- RandomRotate, RGB2Gray, AdjustGamma, Rerange, CLAHE -- are outside SURVEY section 8f rank 4.)"""
import copy

import numpy as np
import pytest


def _pipelines():
    import segdistill_amd
    from segdistill_amd.data.pipelines import PIPELINES
    from segdistill_amd.registry import build_from_cfg
    segdistill_amd.register_all()
    return lambda cfg: build_from_cfg(dict(cfg), PIPELINES)


def _results(with_seg=False):
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (288, 512, 3), dtype=np.uint8)          # the shape of the reference's color.jpg (test_transform.py:39)
    res = dict(img=img, img_shape=img.shape, ori_shape=img.shape, pad_shape=img.shape, scale_factor=1.0)
    if with_seg:
        res['gt_semantic_seg'] = rng.randint(0, 19, (288, 512)).astype(np.uint8)
        res['seg_fields'] = ['gt_semantic_seg']
    return res


def test_resize_reference_assertions():
    build = _pipelines()
    with pytest.raises(AssertionError):                                # img_scale given as a list of ints
        build(dict(type='Resize', img_scale=[1333, 800], keep_ratio=True))
    with pytest.raises(AssertionError):                                # several scales together with a ratio range
        build(dict(type='Resize', img_scale=[(1333, 800), (1333, 600)], ratio_range=(0.9, 1.1), keep_ratio=True))
    with pytest.raises(AssertionError):                                # unknown multiscale mode
        build(dict(type='Resize', img_scale=[(1333, 800), (1333, 600)], keep_ratio=True, multiscale_mode='2333'))
    results = _results()
    out = build(dict(type='Resize', img_scale=(1333, 800), keep_ratio=True))(copy.deepcopy(results))
    assert out['img_shape'] == (750, 1333, 3)
    out = build(dict(type='Resize', img_scale=(1280, 800), multiscale_mode='value', keep_ratio=False))(copy.deepcopy(results))
    assert out['img_shape'] == (800, 1280, 3)
    out = build(dict(type='Resize', img_scale=[(1333, 400), (1333, 1200)], multiscale_mode='range', keep_ratio=True))(copy.deepcopy(results))
    assert max(out['img_shape'][:2]) <= 1333 and 400 <= min(out['img_shape'][:2]) <= 1200
    out = build(dict(type='Resize', img_scale=[(1333, 800), (1333, 400)], multiscale_mode='value', keep_ratio=True))(copy.deepcopy(results))
    assert out['img_shape'] in [(750, 1333, 3), (400, 711, 3)]
    out = build(dict(type='Resize', img_scale=(1333, 800), ratio_range=(0.9, 1.1), keep_ratio=True))(copy.deepcopy(results))
    assert max(out['img_shape'][:2]) <= 1333 * 1.1
    out = build(dict(type='Resize', img_scale=None, ratio_range=(0.5, 2.0), keep_ratio=True))(copy.deepcopy(results))
    assert int(288 * 0.5) <= out['img_shape'][0] <= 288 * 2.0 and int(512 * 0.5) <= out['img_shape'][1] <= 512 * 2.0


def test_flip_reference_assertions():
    build = _pipelines()
    with pytest.raises(AssertionError):
        build(dict(type='RandomFlip', prob=1.5))
    with pytest.raises(AssertionError):
        build(dict(type='RandomFlip', prob=1, direction='horizonta'))
    results = _results(with_seg=True)
    img0, seg0 = results['img'].copy(), results['gt_semantic_seg'].copy()
    once = build(dict(type='RandomFlip', prob=1))(results)
    assert not np.array_equal(once['img'], img0)
    once.pop('flip'), once.pop('flip_direction')                      # the reference builds a fresh module and flips again (:139-140)
    twice = build(dict(type='RandomFlip', prob=1))(once)
    assert np.array_equal(twice['img'], img0) and np.array_equal(twice['gt_semantic_seg'], seg0)


def test_random_crop_reference_assertions():
    build = _pipelines()
    with pytest.raises(AssertionError):
        build(dict(type='RandomCrop', crop_size=(-1, 0)))
    results = _results(with_seg=True)
    h, w, _ = results['img'].shape
    out = build(dict(type='RandomCrop', crop_size=(h - 20, w - 20)))(results)
    assert out['img'].shape[:2] == (h - 20, w - 20) and out['img_shape'][:2] == (h - 20, w - 20)
    assert out['gt_semantic_seg'].shape[:2] == (h - 20, w - 20)


def test_pad_reference_assertions():
    build = _pipelines()
    with pytest.raises(AssertionError):
        build(dict(type='Pad'))
    pad = build(dict(type='Pad', size_divisor=32))
    results = _results()
    img0 = results['img'].copy()
    out = pad(results)
    assert np.array_equal(out['img'], img0)                            # 288 x 512 is already divisible by 32
    assert out['img'].shape[0] % 32 == 0 and out['img'].shape[1] % 32 == 0
    out = pad(build(dict(type='Resize', img_scale=(1333, 800), keep_ratio=True))(out))
    assert out['img'].shape[0] % 32 == 0 and out['img'].shape[1] % 32 == 0


def test_normalize_reference_formula():
    build = _pipelines()
    cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    results = _results()
    img0 = results['img'].copy()
    out = build(dict(type='Normalize', **cfg))(results)
    expected = (img0[..., ::-1] - np.array(cfg['mean'])) / np.array(cfg['std'])
    assert np.allclose(out['img'], expected, atol=1e-5)


def test_seg_rescale_reference_assertions():
    build = _pipelines()
    results = _results(with_seg=True)
    results = dict(gt_semantic_seg=results['gt_semantic_seg'], seg_fields=['gt_semantic_seg'])
    h, w = results['gt_semantic_seg'].shape
    out = build(dict(type='SegRescale', scale_factor=1. / 2))(copy.deepcopy(results))
    assert out['gt_semantic_seg'].shape == (h // 2, w // 2)
    out = build(dict(type='SegRescale', scale_factor=1))(copy.deepcopy(results))
    assert out['gt_semantic_seg'].shape == (h, w)

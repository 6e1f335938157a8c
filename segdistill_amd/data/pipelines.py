"""Training pipeline transforms of the reference (mmseg/datasets/pipelines/{loading,transforms,formating,compose}.py), the
ones `local_configs/_base_/datasets/ade20k_repeat.py:7-18` uses, on a `results` dict with the reference's keys.  Random
draws use the global numpy RNG in the reference's order, so a seeded run makes the same decisions."""
from __future__ import annotations

import os.path as osp

import numpy as np
import torch

from ..registry import Registry, build_from_cfg
from . import imops

PIPELINES = Registry('pipeline')


class Compose:
    """compose.py:9-51."""

    def __init__(self, transforms):
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]
        # a Resize whose result goes straight into a RandomCrop leaves the image resize to the crop (imops.LazyResize: only the window is computed)
        for a, b in zip(self.transforms, self.transforms[1:]):
            if isinstance(a, Resize) and isinstance(b, RandomCrop):
                a.crop_follows = True

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


def _imread(path, unchanged=False):
    from PIL import Image
    with Image.open(path) as im:
        if unchanged:
            return np.array(im)
        return np.array(im.convert('RGB'))[:, :, ::-1].copy()   # BGR, as cv2.imread / mmcv 'color'


@PIPELINES.register_module()
class LoadImageFromFile:
    """loading.py:10-86: uint8 BGR image + the default meta keys."""

    def __init__(self, to_float32=False, color_type='color', **kwargs):
        self.to_float32 = to_float32

    def __call__(self, results):
        name = results['img_info']['filename']
        filename = osp.join(results['img_prefix'], name) if results.get('img_prefix') is not None else name
        img = _imread(filename)
        if self.to_float32:
            img = img.astype(np.float32)
        results.update(filename=filename, ori_filename=name, img=img, img_shape=img.shape, ori_shape=img.shape, pad_shape=img.shape,
                       scale_factor=1.0,
                       img_norm_cfg=dict(mean=np.zeros(3, np.float32), std=np.ones(3, np.float32), to_rgb=False))
        return results


@PIPELINES.register_module()
class LoadAnnotations:
    """loading.py:89-153.  reduce_zero_label: 0 -> 255, then every label - 1, 254 -> 255 (:141-145)."""

    def __init__(self, reduce_zero_label=False, **kwargs):
        self.reduce_zero_label = reduce_zero_label

    def __call__(self, results):
        name = results['ann_info']['seg_map']
        filename = osp.join(results['seg_prefix'], name) if results.get('seg_prefix') is not None else name
        seg = _imread(filename, unchanged=True).squeeze().astype(np.uint8)
        if results.get('label_map') is not None:
            for old, new in results['label_map'].items():
                seg[seg == old] = new
        if self.reduce_zero_label:
            seg[seg == 0] = 255
            seg = seg - 1
            seg[seg == 254] = 255
        results['gt_semantic_seg'] = seg
        results.setdefault('seg_fields', []).append('gt_semantic_seg')
        return results


@PIPELINES.register_module()
class Resize:
    """transforms.py:237-463 (img_scale / multiscale_mode / ratio_range / keep_ratio)."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True):
        if img_scale is None:
            self.img_scale = None
        else:
            self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
            assert all(isinstance(s, tuple) for s in self.img_scale)      # mmcv.is_list_of(img_scale, tuple), transforms.py:264
        if ratio_range is not None:
            assert self.img_scale is None or len(self.img_scale) == 1
        else:
            assert multiscale_mode in ('value', 'range')
        self.multiscale_mode, self.ratio_range, self.keep_ratio = multiscale_mode, ratio_range, keep_ratio

    def _random_scale(self, results):
        if self.ratio_range is not None:
            base = self.img_scale[0] if self.img_scale is not None else results['img'].shape[:2][::-1]
            lo, hi = self.ratio_range
            ratio = np.random.random_sample() * (hi - lo) + lo
            scale, idx = (int(base[0] * ratio), int(base[1] * ratio)), None
        elif len(self.img_scale) == 1:
            scale, idx = self.img_scale[0], 0
        elif self.multiscale_mode == 'range':
            longs, shorts = [max(s) for s in self.img_scale], [min(s) for s in self.img_scale]
            scale = (np.random.randint(min(longs), max(longs) + 1), np.random.randint(min(shorts), max(shorts) + 1))
            idx = None
        else:
            idx = np.random.randint(len(self.img_scale))
            scale = self.img_scale[idx]
        results['scale'], results['scale_idx'] = scale, idx

    def __call__(self, results):
        if 'scale' not in results:
            self._random_scale(results)
        img = results['img']
        h, w = img.shape[:2]
        if getattr(self, 'crop_follows', False) and img.dtype == np.uint8:
            size = imops.rescale_size((w, h), results['scale']) if self.keep_ratio else results['scale']
            new = imops.LazyResize(img, size)                 # computed by the RandomCrop that follows, for its window only
        elif self.keep_ratio:
            new = imops.imrescale(img, results['scale'])
        else:
            new = imops.imresize(img, results['scale'])
        nh, nw = new.shape[:2]
        results.update(img=new, img_shape=new.shape, pad_shape=new.shape, keep_ratio=self.keep_ratio,
                       scale_factor=np.array([nw / w, nh / h, nw / w, nh / h], dtype=np.float32))
        for key in results.get('seg_fields', []):
            if self.keep_ratio:
                results[key] = imops.imrescale(results[key], results['scale'], interpolation='nearest')
            else:
                results[key] = imops.imresize(results[key], results['scale'], interpolation='nearest')
        return results


@PIPELINES.register_module()
class RandomCrop:
    """transforms.py:724-793: up to 10 re-draws until no class covers more than cat_max_ratio of the crop."""

    def __init__(self, crop_size, cat_max_ratio=1., ignore_index=255):
        assert crop_size[0] > 0 and crop_size[1] > 0
        self.crop_size, self.cat_max_ratio, self.ignore_index = crop_size, cat_max_ratio, ignore_index

    def get_crop_bbox(self, img):
        mh, mw = max(img.shape[0] - self.crop_size[0], 0), max(img.shape[1] - self.crop_size[1], 0)
        oh, ow = np.random.randint(0, mh + 1), np.random.randint(0, mw + 1)
        return oh, oh + self.crop_size[0], ow, ow + self.crop_size[1]

    @staticmethod
    def crop(img, box):
        y1, y2, x1, x2 = box
        if isinstance(img, imops.LazyResize):
            return img.region(y1, y2, x1, x2)
        return img[y1:y2, x1:x2, ...]

    def __call__(self, results):
        img = results['img']
        box = self.get_crop_bbox(img)
        if self.cat_max_ratio < 1.:
            for _ in range(10):
                seg = self.crop(results['gt_semantic_seg'], box)
                if seg.dtype == np.uint8:          # the counts np.unique returns, without its sort of 262 144 labels
                    full = np.bincount(seg.reshape(-1), minlength=256)
                    labels = np.nonzero(full)[0]
                    cnt = full[labels]
                else:
                    labels, cnt = np.unique(seg, return_counts=True)
                cnt = cnt[labels != self.ignore_index]
                if len(cnt) > 1 and np.max(cnt) / np.sum(cnt) < self.cat_max_ratio:
                    break
                box = self.get_crop_bbox(img)
        img = self.crop(img, box)
        results['img'], results['img_shape'] = img, img.shape
        for key in results.get('seg_fields', []):
            results[key] = self.crop(results[key], box)
        return results


@PIPELINES.register_module()
class RandomFlip:
    """transforms.py:465-517."""

    def __init__(self, prob=None, direction='horizontal'):
        if prob is not None:
            assert 0 <= prob <= 1
        assert direction in ('horizontal', 'vertical')
        self.prob, self.direction = prob, direction

    def __call__(self, results):
        if 'flip' not in results:
            results['flip'] = bool(np.random.rand() < self.prob)
        results.setdefault('flip_direction', self.direction)
        if results['flip']:
            results['img'] = imops.imflip(results['img'], results['flip_direction'])
            for key in results.get('seg_fields', []):
                results[key] = imops.imflip(results[key], results['flip_direction']).copy()
        return results


@PIPELINES.register_module()
class PhotoMetricDistortion:
    """transforms.py:1099-1214: brightness, contrast (first or last), saturation, hue -- each with probability 1/2."""

    def __init__(self, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
        self.brightness_delta = brightness_delta
        self.contrast_lower, self.contrast_upper = contrast_range
        self.saturation_lower, self.saturation_upper = saturation_range
        self.hue_delta = hue_delta

    @staticmethod
    def convert(img, alpha=1, beta=0):
        """transforms.py:1128-1131: float32 multiply-add, clip, cast back (truncation).  For 8-bit input the 256 possible results are a table: the
        same arithmetic per value, one gather per pixel."""
        if img.dtype == np.uint8:
            lut = np.clip(np.arange(256, dtype=np.float32) * alpha + beta, 0, 255).astype(np.uint8)
            return lut[img]
        return np.clip(img.astype(np.float32) * alpha + beta, 0, 255).astype(np.uint8)

    def brightness(self, img):
        if np.random.randint(2):
            return self.convert(img, beta=np.random.uniform(-self.brightness_delta, self.brightness_delta))
        return img

    def contrast(self, img):
        if np.random.randint(2):
            return self.convert(img, alpha=np.random.uniform(self.contrast_lower, self.contrast_upper))
        return img

    def saturation(self, img):
        if np.random.randint(2):
            hsv = imops.bgr2hsv(img)
            hsv[:, :, 1] = self.convert(hsv[:, :, 1], alpha=np.random.uniform(self.saturation_lower, self.saturation_upper))
            img = imops.hsv2bgr(hsv)
        return img

    def hue(self, img):
        if np.random.randint(2):
            hsv = imops.bgr2hsv(img)
            hsv[:, :, 0] = (hsv[:, :, 0].astype(int) + np.random.randint(-self.hue_delta, self.hue_delta)) % 180
            img = imops.hsv2bgr(hsv)
        return img

    def __call__(self, results):
        img = self.brightness(results['img'])
        mode = np.random.randint(2)
        if mode == 1:
            img = self.contrast(img)
        img = self.hue(self.saturation(img))
        if mode == 0:
            img = self.contrast(img)
        results['img'] = img
        return results


@PIPELINES.register_module()
class Normalize:
    """transforms.py:591-630."""

    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std, self.to_rgb = np.array(mean, np.float32), np.array(std, np.float32), to_rgb

    def __call__(self, results):
        results['img'] = imops.imnormalize(results['img'], self.mean, self.std, self.to_rgb)
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results


@PIPELINES.register_module()
class Pad:
    """transforms.py:520-588: to a fixed size or up to a multiple of size_divisor (bottom / right)."""

    def __init__(self, size=None, size_divisor=None, pad_val=0, seg_pad_val=255):
        assert (size is None) != (size_divisor is None)
        self.size, self.size_divisor, self.pad_val, self.seg_pad_val = size, size_divisor, pad_val, seg_pad_val

    def __call__(self, results):
        img = results['img']
        if self.size is not None:
            shape = tuple(self.size)
        else:
            d = self.size_divisor
            shape = (int(np.ceil(img.shape[0] / d)) * d, int(np.ceil(img.shape[1] / d)) * d)
        img = imops.impad(img, shape, self.pad_val)
        results.update(img=img, pad_shape=img.shape, pad_fixed_size=self.size, pad_size_divisor=self.size_divisor)
        for key in results.get('seg_fields', []):
            results[key] = imops.impad(results[key], img.shape[:2], self.seg_pad_val)
        return results


@PIPELINES.register_module()
class SegRescale:
    """transforms.py:1070-1096: rescale the label maps only (nearest), e.g. to the stride of an auxiliary output."""

    def __init__(self, scale_factor=1):
        self.scale_factor = scale_factor

    def __call__(self, results):
        for key in results.get('seg_fields', []):
            if self.scale_factor != 1:
                seg = results[key]
                h, w = seg.shape[:2]
                results[key] = imops.imresize(seg, (int(w * float(self.scale_factor) + 0.5), int(h * float(self.scale_factor) + 0.5)), 'nearest')
        return results


@PIPELINES.register_module()
class DefaultFormatBundle:
    """formating.py:181-221: img -> float CHW tensor, gt_semantic_seg -> int64 [1,H,W] tensor (plain tensors, no DataContainer)."""

    def __call__(self, results):
        if 'img' in results:
            img = results['img']
            if img.ndim < 3:
                img = np.expand_dims(img, -1)
            results['img'] = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        if 'gt_semantic_seg' in results:
            results['gt_semantic_seg'] = torch.from_numpy(results['gt_semantic_seg'][None, ...].astype(np.int64))
        return results


@PIPELINES.register_module()
class Collect:
    """formating.py:224-283."""

    META = ('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip', 'flip_direction', 'img_norm_cfg')

    def __init__(self, keys, meta_keys=META):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {k: results[k] for k in self.keys}
        data['img_metas'] = {k: results[k] for k in self.meta_keys if k in results}
        return data

"""Directory datasets + loader of the reference (mmseg/datasets/{custom,ade,dataset_wrappers,builder}.py), train path."""
from __future__ import annotations

import os
import os.path as osp
import random
from functools import partial

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, DistributedSampler

from ..registry import Registry, build_from_cfg
from .pipelines import Compose

DATASETS = Registry('dataset')


def _scandir(root, suffix):
    """mmcv.scandir(root, suffix, recursive=True): relative paths, sorted for a run-to-run stable order."""
    out = []
    for base, _, files in os.walk(root):
        for f in files:
            if f.endswith(suffix):
                out.append(osp.relpath(osp.join(base, f), root))
    return sorted(out)


@DATASETS.register_module()
class CustomDataset(Dataset):
    """custom.py:16-208 (train path): images in img_dir, label maps in ann_dir, same stem, different suffix; optional split file."""

    CLASSES = None
    PALETTE = None

    def __init__(self, pipeline, img_dir, img_suffix='.jpg', ann_dir=None, seg_map_suffix='.png', split=None, data_root=None,
                 test_mode=False, ignore_index=255, reduce_zero_label=False, classes=None, palette=None):
        self.pipeline = Compose(pipeline)
        self.img_dir, self.img_suffix, self.ann_dir, self.seg_map_suffix = img_dir, img_suffix, ann_dir, seg_map_suffix
        self.split, self.data_root, self.test_mode = split, data_root, test_mode
        self.ignore_index, self.reduce_zero_label = ignore_index, reduce_zero_label
        self.label_map, self.custom_classes = None, False
        if classes is not None:
            self.CLASSES = tuple(classes)
        if self.data_root is not None:
            if not osp.isabs(self.img_dir):
                self.img_dir = osp.join(self.data_root, self.img_dir)
            if not (self.ann_dir is None or osp.isabs(self.ann_dir)):
                self.ann_dir = osp.join(self.data_root, self.ann_dir)
            if not (self.split is None or osp.isabs(self.split)):
                self.split = osp.join(self.data_root, self.split)
        self.img_infos = self.load_annotations(self.img_dir, self.img_suffix, self.ann_dir, self.seg_map_suffix, self.split)

    def __len__(self):
        return len(self.img_infos)

    def load_annotations(self, img_dir, img_suffix, ann_dir, seg_map_suffix, split):
        infos = []
        if split is not None:
            with open(split) as f:
                for line in f:
                    name = line.strip()
                    info = dict(filename=name + img_suffix)
                    if ann_dir is not None:
                        info['ann'] = dict(seg_map=name + seg_map_suffix)
                    infos.append(info)
        else:
            if not osp.isdir(img_dir):
                raise FileNotFoundError(f'image directory {img_dir!r} not found')
            for img in _scandir(img_dir, img_suffix):
                info = dict(filename=img)
                if ann_dir is not None:
                    info['ann'] = dict(seg_map=img.replace(img_suffix, seg_map_suffix))
                infos.append(info)
        return infos

    def get_ann_info(self, idx):
        return self.img_infos[idx]['ann']

    def pre_pipeline(self, results):
        results['seg_fields'] = []
        results['img_prefix'] = self.img_dir
        results['seg_prefix'] = self.ann_dir
        if self.custom_classes:
            results['label_map'] = self.label_map

    def __getitem__(self, idx):
        info = self.img_infos[idx]
        results = dict(img_info=info) if self.test_mode else dict(img_info=info, ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)


@DATASETS.register_module()
class ADE20KDataset(CustomDataset):
    """ade.py:6-84: 150 classes, '.jpg' / '.png', label 0 = "other" is ignored (reduce_zero_label=True)."""

    NUM_CLASSES = 150

    def __init__(self, **kwargs):
        kwargs.setdefault('img_suffix', '.jpg')
        kwargs.setdefault('seg_map_suffix', '.png')
        kwargs.setdefault('reduce_zero_label', True)
        super().__init__(**kwargs)


class RepeatDataset(Dataset):
    """dataset_wrappers.py:21-50."""

    def __init__(self, dataset, times):
        self.dataset, self.times = dataset, times
        self.CLASSES = getattr(dataset, 'CLASSES', None)
        self._ori_len = len(dataset)

    def __getitem__(self, idx):
        return self.dataset[idx % self._ori_len]

    def __len__(self):
        return self.times * self._ori_len


def build_dataset(cfg, default_args=None):
    """builder.py:60-74 (RepeatDataset / lists of configs / plain datasets)."""
    cfg = dict(cfg)
    if cfg.get('type') == 'RepeatDataset':
        return RepeatDataset(build_dataset(cfg['dataset'], default_args), cfg['times'])
    return build_from_cfg(cfg, DATASETS, default_args)


def worker_init_fn(worker_id, num_workers, rank, seed):
    """builder.py:152-169: every (rank, worker) gets its own numpy / python seed."""
    s = num_workers * rank + worker_id + seed
    np.random.seed(s)
    random.seed(s)


def collate(batch):
    """Stack the fixed-size crops; metas stay a list (the KD path never reads them)."""
    out = {'img': torch.stack([b['img'] for b in batch]), 'img_metas': [b.get('img_metas') for b in batch]}
    if 'gt_semantic_seg' in batch[0]:
        out['gt_semantic_seg'] = torch.stack([b['gt_semantic_seg'] for b in batch])
    return out


def build_dataloader(dataset, samples_per_gpu, workers_per_gpu, world=1, rank=0, shuffle=True, seed=None, drop_last=True, pin_memory=True):
    """builder.py:77-149, distributed form: one loader per rank, DistributedSampler shards the (repeated) dataset."""
    sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=seed or 0) if world > 1 else None
    init = partial(worker_init_fn, num_workers=workers_per_gpu, rank=rank, seed=seed) if seed is not None else None
    gen = None
    if seed is not None and sampler is None:
        gen = torch.Generator()
        gen.manual_seed(seed)
    return DataLoader(dataset, batch_size=samples_per_gpu, sampler=sampler, shuffle=(shuffle and sampler is None), num_workers=workers_per_gpu,
                      collate_fn=collate, pin_memory=pin_memory and torch.cuda.is_available(), drop_last=drop_last, worker_init_fn=init,
                      generator=gen, persistent_workers=workers_per_gpu > 0)

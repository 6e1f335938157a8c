"""Registries and build functions with the names the reference's configs use
(reference mmseg/models/builder.py:6-66)."""
from __future__ import annotations

import warnings

from torch import nn

from .registry import Registry, build_from_cfg

BACKBONES = Registry('backbone')
NECKS = Registry('neck')
HEADS = Registry('head')
LOSSES = Registry('loss')
SEGMENTORS = Registry('segmentor')
DISTILL_LOSSES = Registry('distillation loss')  # replaces eval(loss_name) at reference opts.py:83


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_segmentor(cfg, train_cfg=None, test_cfg=None):
    if train_cfg is not None or test_cfg is not None:
        warnings.warn('passing train_cfg / test_cfg beside the model config is the old calling convention: put them inside the model dict', UserWarning)
    assert cfg.get('train_cfg') is None or train_cfg is None, 'train_cfg was given twice: as an argument and inside the model config'
    assert cfg.get('test_cfg') is None or test_cfg is None, 'test_cfg was given twice: as an argument and inside the model config'
    return build(cfg, SEGMENTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))

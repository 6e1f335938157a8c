"""Minimal behavioural stand-ins for the THIRD-PARTY packages the reference imports but
which are not installed in the build container: mmcv-full==1.2.2 (README.md:13),
timm==0.3.2 (requirements.txt:4) and IPython.  TEST INFRASTRUCTURE, build container only.

They exist so that oracle/gen_golden_nets.py can import the reference's network /
segmentor sources unmodified and record their outputs.  Behaviour restated from the
published semantics of those packages; it is therefore "unpinned"
except through parameter-count and state-dict-key agreement with the reference's
published model sizes.  SyncBN is mapped to BatchNorm2d exactly as the reference's own
tests do (tests/test_models/test_forward.py:186-205).
"""
from __future__ import annotations

import inspect
import logging
import math
import sys
import types
import warnings

import torch
import torch.nn as nn


# ------------------------------------------------------------------ mmcv.utils
class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop('type')
    if isinstance(typ, str):
        cls = registry.get(typ)
        if cls is None:
            raise KeyError(f'{typ} is not in the {registry._name} registry')
    else:
        cls = typ
    return cls(**args)


# ------------------------------------------------------------------ mmcv.cnn
def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if hasattr(module, 'weight') and module.weight is not None:
        if distribution == 'uniform':
            nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.normal_(module.weight, mean, std)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


_NORMS = {'BN': ('bn', nn.BatchNorm2d), 'SyncBN': ('bn', nn.BatchNorm2d), 'BN2d': ('bn', nn.BatchNorm2d),
          'GN': ('gn', nn.GroupNorm), 'LN': ('ln', nn.LayerNorm)}


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    typ = cfg.pop('type')
    abbr, cls = _NORMS[typ]
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    if typ == 'GN':
        layer = cls(num_channels=num_features, **cfg)
    else:
        layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


def build_conv_layer(cfg, *args, **kwargs):
    if cfg is not None and cfg.get('type', 'Conv2d') not in ('Conv2d', 'Conv'):
        raise KeyError(cfg)
    return nn.Conv2d(*args, **kwargs)


def build_plugin_layer(*a, **k):
    raise NotImplementedError('plugins are outside the KD path')


class ConvModule(nn.Module):
    """conv -> norm -> act with children named conv / bn|gn / activate; bias = (norm is None)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True, with_spectral_norm=False,
                 padding_mode='zeros', order=('conv', 'norm', 'act')):
        super().__init__()
        assert order == ('conv', 'norm', 'act')
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                              groups=groups, bias=bias)
        self.in_channels, self.out_channels = in_channels, out_channels
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        kaiming_init(self.conv, a=0, nonlinearity='relu')
        if self.with_norm:
            constant_init(getattr(self, self.norm_name), 1, bias=0)

    @property
    def norm(self):
        return getattr(self, self.norm_name)

    def forward(self, x, activate=True, norm=True):
        x = self.conv(x)
        if norm and self.with_norm:
            x = self.norm(x)
        if activate and self.with_activation:
            x = self.activate(x)
        return x


class DepthwiseSeparableConvModule(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError


# ------------------------------------------------------------------ mmcv.runner
def _identity_decorator(*dargs, **dkw):
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]

    def deco(fn):
        return fn
    return deco


# ------------------------------------------------------------------ timm
class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x.div(keep) * mask


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def install():
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_segdistill_stub', False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        parent, _, child = name.rpartition('.')
        if parent:
            setattr(sys.modules[parent], child, m)
        return m

    get_logger = lambda name, **k: logging.getLogger(name)  # noqa: E731
    print_log = lambda msg, logger=None, level=logging.INFO: None  # noqa: E731
    mmcv = mod('mmcv', _segdistill_stub=True, __version__='1.2.2')
    mod('mmcv.utils', Registry=Registry, build_from_cfg=build_from_cfg, get_logger=get_logger, print_log=print_log)
    mod('mmcv.utils.parrots_wrapper', _BatchNorm=nn.modules.batchnorm._BatchNorm, SyncBatchNorm=nn.SyncBatchNorm)
    mod('mmcv.cnn', ConvModule=ConvModule, DepthwiseSeparableConvModule=DepthwiseSeparableConvModule,
        build_norm_layer=build_norm_layer, build_conv_layer=build_conv_layer, build_plugin_layer=build_plugin_layer,
        kaiming_init=kaiming_init, constant_init=constant_init, normal_init=normal_init)
    mod('mmcv.runner', auto_fp16=_identity_decorator, force_fp32=_identity_decorator,
        load_checkpoint=lambda *a, **k: None)
    mmcv.imread = None
    mod('timm')
    mod('timm.models')
    mod('timm.models.layers', DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)
    mod('timm.models.registry', register_model=lambda f: f)
    mod('timm.models.vision_transformer', _cfg=lambda **k: dict(k))
    if 'IPython' not in sys.modules:
        mod('IPython', embed=lambda *a, **k: None)

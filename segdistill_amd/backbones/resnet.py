"""ResNet / ResNetV1c / ResNetV1d with per-stage stride and dilation (the "-d8" encoders of
PSPNet) on PyTorch-ROCm.

Behavioural counterpart of reference mmseg/models/backbones/resnet.py (BasicBlock :13-94,
Bottleneck :97-302, ResNet :305-659, ResNetV1c :662-675, ResNetV1d :678-688) and
mmseg/models/utils/res_layer.py (ResLayer :5-94).  State-dict keys match the reference:
``stem.{0,1,3,4,6,7}`` (deep stem) or ``conv1/bn1``, ``layerN.i.{conv1,bn1,conv2,bn2[,conv3,bn3]}``,
``layerN.0.downsample.{0,1}`` (``{1,2}`` behind the AvgPool of V1d).
Out of scope and rejected loudly: DCN, plugins, gradient checkpointing.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from ..builder import BACKBONES
from ..layers import build_conv_layer, build_norm_layer, constant_init, kaiming_init


class _Residual(nn.Module):
    """Shared plumbing: named norm children, identity / projected shortcut, final ReLU."""

    expansion = 1

    def _add_norm(self, idx, norm_cfg, channels):
        name, layer = build_norm_layer(norm_cfg, channels, postfix=idx)
        self.add_module(name, layer)
        setattr(self, f'norm{idx}_name', name)

    def _norm(self, idx):
        return getattr(self, getattr(self, f'norm{idx}_name'))

    norm1 = property(lambda self: self._norm(1))
    norm2 = property(lambda self: self._norm(2))
    norm3 = property(lambda self: self._norm(3))

    def forward(self, x):
        if self._frozen_fast(x):
            return self._forward_frozen(x)
        shortcut = x if self.downsample is None else self.downsample(x)
        return self.relu(self.body(x) + shortcut)

    # ---- frozen network (the teacher): conv -> [eval BatchNorm (+ shortcut) + ReLU as ONE in-place pass] (csrc/affine_act.hip) ----
    def _pairs(self):
        n = 3 if hasattr(self, 'conv3') else 2
        return [(getattr(self, f'conv{i}'), self._norm(i)) for i in range(1, n + 1)]

    def _frozen_fast(self, x):
        from .. import affine_act
        if self.training or torch.is_grad_enabled() or not (x.is_cuda and x.dtype == torch.float32):
            return False
        mods = [self, self.relu] + [m for pair in self._pairs() for m in pair]
        return affine_act._ENABLED and not any(m._forward_hooks or m._forward_pre_hooks for m in mods)

    def _forward_frozen(self, x):
        from .. import affine_act
        shortcut = x if self.downsample is None else affine_act.run_frozen_sequential(self.downsample, x)
        pairs = self._pairs()
        y = x
        for i, (conv, norm) in enumerate(pairs):
            y = conv(y)
            last = i == len(pairs) - 1
            if affine_act.usable(y, norm):
                y = affine_act.eval_norm_act_(y, norm, True, shortcut if last else None)
            else:
                y = norm(y)
                y = self.relu(y + shortcut) if last else self.relu(y)
        return y


class BasicBlock(_Residual):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', conv_cfg=None,
                 norm_cfg=dict(type='BN')):
        super().__init__()
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self._add_norm(1, norm_cfg, planes)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self._add_norm(2, norm_cfg, planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride, self.dilation = stride, dilation

    def body(self, x):
        x = self.relu(self.norm1(self.conv1(x)))
        return self.norm2(self.conv2(x))


class Bottleneck(_Residual):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch', conv_cfg=None,
                 norm_cfg=dict(type='BN')):
        super().__init__()
        if style not in ('pytorch', 'caffe'):
            raise AssertionError(style)
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)  # which conv carries the stride
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 1, stride=s1, bias=False)
        self._add_norm(1, norm_cfg, planes)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)
        self._add_norm(2, norm_cfg, planes)
        self.conv3 = build_conv_layer(conv_cfg, planes, planes * self.expansion, 1, bias=False)
        self._add_norm(3, norm_cfg, planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride, self.dilation = stride, dilation

    def body(self, x):
        x = self.relu(self.norm1(self.conv1(x)))
        x = self.relu(self.norm2(self.conv2(x)))
        return self.norm3(self.conv3(x))


class ResLayer(nn.Sequential):
    """One stage: first block carries stride + shortcut projection, the rest are stride 1."""

    def __init__(self, block, inplanes, planes, num_blocks, stride=1, dilation=1, avg_down=False, conv_cfg=None,
                 norm_cfg=dict(type='BN'), multi_grid=None, contract_dilation=False, **block_kw):
        out_ch = planes * block.expansion
        shortcut = None
        if stride != 1 or inplanes != out_ch:
            mods = []
            conv_stride = stride
            if avg_down:
                conv_stride = 1
                mods.append(nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True, count_include_pad=False))
            mods += [build_conv_layer(conv_cfg, inplanes, out_ch, kernel_size=1, stride=conv_stride, bias=False),
                     build_norm_layer(norm_cfg, out_ch)[1]]
            shortcut = nn.Sequential(*mods)
        if multi_grid is not None:
            first = multi_grid[0]
        else:
            first = dilation // 2 if (dilation > 1 and contract_dilation) else dilation
        blocks = [block(inplanes=inplanes, planes=planes, stride=stride, dilation=first, downsample=shortcut, conv_cfg=conv_cfg,
                        norm_cfg=norm_cfg, **block_kw)]
        for i in range(1, num_blocks):
            blocks.append(block(inplanes=out_ch, planes=planes, stride=1, dilation=dilation if multi_grid is None else multi_grid[i],
                                conv_cfg=conv_cfg, norm_cfg=norm_cfg, **block_kw))
        super().__init__(*blocks)
        self.block = block


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
                     101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}

    def __init__(self, depth, in_channels=3, stem_channels=64, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch', deep_stem=False, avg_down=False,
                 frozen_stages=-1, conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True), norm_eval=False, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, multi_grid=None, contract_dilation=False,
                 with_cp=False, zero_init_residual=True):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet')
        if dcn is not None or plugins is not None or with_cp:
            raise NotImplementedError('DCN / plugins / checkpointing are outside the KD path')
        assert 1 <= num_stages <= 4 and len(strides) == len(dilations) == num_stages and max(out_indices) < num_stages
        self.depth, self.out_indices = depth, tuple(out_indices)
        self.deep_stem, self.avg_down = deep_stem, avg_down
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg
        self.zero_init_residual = zero_init_residual
        self.block, counts = self.arch_settings[depth]
        self.stage_blocks = counts[:num_stages]
        self._build_stem(in_channels, stem_channels)
        inplanes = stem_channels
        self.res_layers = []
        for i, n in enumerate(self.stage_blocks):
            planes = base_channels * 2 ** i
            stage = ResLayer(self.block, inplanes, planes, n, stride=strides[i], dilation=dilations[i], avg_down=avg_down,
                             conv_cfg=conv_cfg, norm_cfg=norm_cfg, multi_grid=multi_grid if i == len(self.stage_blocks) - 1 else None,
                             contract_dilation=contract_dilation, style=style)
            inplanes = planes * self.block.expansion
            name = f'layer{i + 1}'
            self.add_module(name, stage)
            self.res_layers.append(name)
        self._freeze_stages()
        self.feat_dim = inplanes

    def _build_stem(self, cin, cstem):
        if self.deep_stem:
            half = cstem // 2
            seq = []
            for a, b, s in ((cin, half, 2), (half, half, 1), (half, cstem, 1)):
                seq += [build_conv_layer(self.conv_cfg, a, b, kernel_size=3, stride=s, padding=1, bias=False),
                        build_norm_layer(self.norm_cfg, b)[1], nn.ReLU(inplace=True)]
            self.stem = nn.Sequential(*seq)
        else:
            self.conv1 = build_conv_layer(self.conv_cfg, cin, cstem, kernel_size=7, stride=2, padding=3, bias=False)
            self.norm1_name, n1 = build_norm_layer(self.norm_cfg, cstem, postfix=1)
            self.add_module(self.norm1_name, n1)
            self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            frozen = [self.stem] if self.deep_stem else [self.conv1, self.norm1]
            for m in frozen:
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
            return
        if pretrained is not None:
            raise TypeError('pretrained must be a str or None')
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                kaiming_init(m)
            elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                constant_init(m, 1)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    constant_init(m.norm3, 0)
                elif isinstance(m, BasicBlock):
                    constant_init(m.norm2, 0)

    def forward(self, x):
        if self.deep_stem and not self.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32:
            from .. import affine_act
            x = affine_act.run_frozen_sequential(self.stem, x)        # frozen network: conv -> [BatchNorm + ReLU in one pass] x 3
        else:
            x = self.stem(x) if self.deep_stem else self.relu(self.norm1(self.conv1(x)))
        x = self.maxpool(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self


@BACKBONES.register_module()
class ResNetV1c(ResNet):
    """Three 3x3 convs in the stem instead of one 7x7."""

    def __init__(self, **kwargs):
        super().__init__(deep_stem=True, avg_down=False, **kwargs)


@BACKBONES.register_module()
class ResNetV1d(ResNet):
    """V1c stem + average-pool down-sampling in the shortcut."""

    def __init__(self, **kwargs):
        super().__init__(deep_stem=True, avg_down=True, **kwargs)

"""UPerNet head (PPM on the coarsest level + FPN top-down fusion); the frozen Swin teacher of
BASELINE config 4 uses it.  Counterpart of reference
mmseg/models/decode_heads/uper_head.py (UPerHead :12-126): children ``psp_modules``,
``bottleneck``, ``lateral_convs.{i}``, ``fpn_convs.{i}``, ``fpn_bottleneck``, ``conv_seg``."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..builder import HEADS
from ..layers import ConvModule, resize
from .decode_head import BaseDecodeHead
from .psp_head import PPM


@HEADS.register_module()
class UPerHead(BaseDecodeHead):
    def __init__(self, pool_scales=(1, 2, 3, 6), **kwargs):
        super().__init__(input_transform='multiple_select', **kwargs)
        cfgs = dict(conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg, act_cfg=self.act_cfg)
        top = self.in_channels[-1]
        self.psp_modules = PPM(pool_scales, top, self.channels, align_corners=self.align_corners, **cfgs)
        self.bottleneck = ConvModule(top + len(pool_scales) * self.channels, self.channels, 3, padding=1, **cfgs)
        self.lateral_convs = nn.ModuleList(ConvModule(c, self.channels, 1, inplace=False, **cfgs) for c in self.in_channels[:-1])
        self.fpn_convs = nn.ModuleList(ConvModule(self.channels, self.channels, 3, padding=1, inplace=False, **cfgs)
                                       for _ in self.in_channels[:-1])
        self.fpn_bottleneck = ConvModule(len(self.in_channels) * self.channels, self.channels, 3, padding=1, **cfgs)

    def psp_forward(self, inputs):
        x = inputs[-1]
        return self.bottleneck(torch.cat([x] + self.psp_modules(x), dim=1))

    def forward(self, inputs):
        inputs = self._transform_inputs(inputs)
        lat = [conv(inputs[i]) for i, conv in enumerate(self.lateral_convs)]
        lat.append(self.psp_forward(inputs))
        for i in range(len(lat) - 1, 0, -1):  # top-down accumulation
            lat[i - 1] = lat[i - 1] + resize(lat[i], size=lat[i - 1].shape[2:], mode='bilinear', align_corners=self.align_corners)
        outs = [self.fpn_convs[i](lat[i]) for i in range(len(lat) - 1)] + [lat[-1]]
        size = outs[0].shape[2:]
        outs = [outs[0]] + [resize(o, size=size, mode='bilinear', align_corners=self.align_corners) for o in outs[1:]]
        return self.cls_seg(self.fpn_bottleneck(torch.cat(outs, dim=1)))

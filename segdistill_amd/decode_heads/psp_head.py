"""Pyramid Pooling head (PSPNet).  Counterpart of reference
mmseg/models/decode_heads/psp_head.py (PPM :10-58, PSPHead :61-101): children
``psp_modules.{i}.{0:AdaptiveAvgPool2d,1:ConvModule}``, ``bottleneck``, ``conv_seg``."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..builder import HEADS
from ..layers import ConvModule, resize
from .decode_head import BaseDecodeHead


class PPM(nn.ModuleList):
    """One (adaptive average pool -> 1x1 ConvModule) branch per pooling scale; forward returns
    the branch outputs bilinearly resized back to the input resolution."""

    def __init__(self, pool_scales, in_channels, channels, conv_cfg, norm_cfg, act_cfg, align_corners):
        super().__init__(
            nn.Sequential(nn.AdaptiveAvgPool2d(s), ConvModule(in_channels, channels, 1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                                              act_cfg=act_cfg)) for s in pool_scales)
        self.pool_scales = pool_scales
        self.align_corners = align_corners
        self.in_channels, self.channels = in_channels, channels

    def forward(self, x):
        size = x.shape[2:]
        from .. import ppm
        scales = [s if isinstance(s, int) else None for s in self.pool_scales]
        # a tapped branch (Extractor hooks on `psp_modules.i`, the reference's child name, or on its pool) must be CALLED as a module
        hooked = any(b._forward_hooks or b._forward_pre_hooks or b[0]._forward_hooks or b[0]._forward_pre_hooks for b in self)
        if None not in scales and ppm.supported(x, scales) and not hooked:
            # every pool scale from ONE read of the map; in the backward dx is gathered once, without float atomics (csrc/ppm_pool.hip)
            pooled = ppm.ppm_pool(x, scales)
            return [resize(branch[1](p), size=size, mode='bilinear', align_corners=self.align_corners) for branch, p in zip(self, pooled)]
        return [resize(branch(x), size=size, mode='bilinear', align_corners=self.align_corners) for branch in self]


@HEADS.register_module()
class PSPHead(BaseDecodeHead):
    def __init__(self, pool_scales=(1, 2, 3, 6), **kwargs):
        super().__init__(**kwargs)
        assert isinstance(pool_scales, (list, tuple))
        self.pool_scales = pool_scales
        cfgs = dict(conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg, act_cfg=self.act_cfg)
        self.psp_modules = PPM(pool_scales, self.in_channels, self.channels, align_corners=self.align_corners, **cfgs)
        self.bottleneck = ConvModule(self.in_channels + len(pool_scales) * self.channels, self.channels, 3, padding=1, **cfgs)

    def forward(self, inputs):
        x = self._transform_inputs(inputs)
        y = self.bottleneck(torch.cat([x] + self.psp_modules(x), dim=1))
        return self.cls_seg(y)

"""FCN head (used as the auxiliary deep-supervision head of PSPNet / UPerNet).
Counterpart of reference mmseg/models/decode_heads/fcn_head.py (FCNHead :10-81):
children ``convs.{i}``, optional ``conv_cat``, ``conv_seg``."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..builder import HEADS
from ..layers import ConvModule
from .decode_head import BaseDecodeHead


@HEADS.register_module()
class FCNHead(BaseDecodeHead):
    def __init__(self, num_convs=2, kernel_size=3, concat_input=True, **kwargs):
        assert num_convs >= 0
        super().__init__(**kwargs)
        self.num_convs, self.kernel_size, self.concat_input = num_convs, kernel_size, concat_input
        if num_convs == 0:
            assert self.in_channels == self.channels
        mk = lambda cin: ConvModule(cin, self.channels, kernel_size=kernel_size, padding=kernel_size // 2,  # noqa: E731
                                    conv_cfg=self.conv_cfg, norm_cfg=self.norm_cfg, act_cfg=self.act_cfg)
        stack = [mk(self.in_channels if i == 0 else self.channels) for i in range(num_convs)]
        self.convs = nn.Sequential(*stack) if stack else nn.Identity()
        if concat_input:
            self.conv_cat = mk(self.in_channels + self.channels)

    def forward(self, inputs):
        x = self._transform_inputs(inputs)
        y = self.convs(x)
        if self.concat_input:
            y = self.conv_cat(torch.cat([x, y], dim=1))
        return self.cls_seg(y)

"""SegFormer all-MLP decode head.

Counterpart of reference mmseg/models/decode_heads/segformer_head.py (MLP :22-33,
SegFormerHead :37-98).  Quirks kept on purpose:
 * the supervised loss is rebuilt with reduction='none' (:45-50), so ``decode.loss_seg`` is a
   [B,H,W] map that ``_parse_losses`` later averages;
 * ``linear_fuse`` always uses SyncBN whatever ``norm_cfg`` says (:66-71);
 * the inherited ``conv_seg`` is never used in forward (Q12) -- it stays a parameter for
   checkpoint-key compatibility and is excluded from gradient reduction by the trainer.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..builder import HEADS, build_loss
from ..layers import ConvModule, resize
from .decode_head import BaseDecodeHead


class MLP(nn.Module):
    """[B,C,h,w] -> tokens [B,h*w,E] through one Linear."""

    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)

    def forward(self, x):
        return self.proj(x.flatten(2).transpose(1, 2))


@HEADS.register_module()
class SegFormerHead(BaseDecodeHead):
    def __init__(self, feature_strides, **kwargs):
        super().__init__(input_transform='multiple_select', **kwargs)
        self.loss_decode = build_loss(dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0, reduction='none'))
        assert len(feature_strides) == len(self.in_channels) and min(feature_strides) == feature_strides[0]
        self.feature_strides = feature_strides
        c1, c2, c3, c4 = self.in_channels
        dim = kwargs['decoder_params']['embed_dim']
        self.linear_c4 = MLP(c4, dim)
        self.linear_c3 = MLP(c3, dim)
        self.linear_c2 = MLP(c2, dim)
        self.linear_c1 = MLP(c1, dim)
        self.linear_fuse = ConvModule(dim * 4, dim, kernel_size=1, norm_cfg=dict(type='SyncBN', requires_grad=True))
        self.linear_pred = nn.Conv2d(dim, self.num_classes, kernel_size=1)

    def forward(self, inputs):
        c1, c2, c3, c4 = self._transform_inputs(inputs)
        n = c1.shape[0]
        size = c1.shape[2:]
        maps = []
        for feat, proj in ((c4, self.linear_c4), (c3, self.linear_c3), (c2, self.linear_c2), (c1, self.linear_c1)):
            m = proj(feat).permute(0, 2, 1).reshape(n, -1, feat.shape[2], feat.shape[3])
            if m.shape[2:] != size:
                m = resize(m, size=size, mode='bilinear', align_corners=False)
            maps.append(m)
        fused = self.linear_fuse(torch.cat(maps, dim=1))
        if self.dropout is not None:
            fused = self.dropout(fused)
        return self.linear_pred(fused)

"""Common part of the segmentation heads: input selection, the 1x1 classifier
``conv_seg``, Dropout2d, and the supervised loss on bilinearly up-sampled logits.

Counterpart of reference mmseg/models/decode_heads/decode_head.py (BaseDecodeHead :14-237:
``_transform_inputs`` :132-158, ``forward_train`` :172-192, ``cls_seg`` :210-215,
``losses`` :217-237).
"""
from __future__ import annotations

from abc import ABCMeta, abstractmethod

import torch
import torch.nn as nn

from ..builder import build_loss
from ..layers import normal_init, resize
from ..losses import accuracy


class BaseDecodeHead(nn.Module, metaclass=ABCMeta):
    def __init__(self, in_channels, channels, *, num_classes, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), in_index=-1, input_transform=None,
                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), decoder_params=None,
                 ignore_index=255, sampler=None, align_corners=False):
        super().__init__()
        self._set_inputs(in_channels, in_index, input_transform)
        self.channels = channels
        self.num_classes = num_classes
        self.dropout_ratio = dropout_ratio
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.loss_decode = build_loss(loss_decode)
        self.ignore_index = ignore_index
        self.align_corners = align_corners
        if sampler is not None:
            raise NotImplementedError('pixel samplers (OHEM) are outside the KD path')
        self.sampler = None
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.fp16_enabled = False

    def extra_repr(self):
        return f'input_transform={self.input_transform}, ignore_index={self.ignore_index}, align_corners={self.align_corners}'

    def _set_inputs(self, in_channels, in_index, input_transform):
        if input_transform not in (None, 'resize_concat', 'multiple_select'):
            raise AssertionError(input_transform)
        self.input_transform, self.in_index = input_transform, in_index
        if input_transform is None:
            assert isinstance(in_channels, int) and isinstance(in_index, int)
            self.in_channels = in_channels
        else:
            assert isinstance(in_channels, (list, tuple)) and isinstance(in_index, (list, tuple))
            assert len(in_channels) == len(in_index)
            self.in_channels = sum(in_channels) if input_transform == 'resize_concat' else in_channels

    def init_weights(self):
        normal_init(self.conv_seg, mean=0, std=0.01)

    def _transform_inputs(self, inputs):
        if self.input_transform == 'resize_concat':
            picked = [inputs[i] for i in self.in_index]
            return torch.cat([resize(x, size=picked[0].shape[2:], mode='bilinear', align_corners=self.align_corners)
                              for x in picked], dim=1)
        if self.input_transform == 'multiple_select':
            return [inputs[i] for i in self.in_index]
        return inputs[self.in_index]

    @abstractmethod
    def forward(self, inputs):
        ...

    def forward_train(self, inputs, img_metas, gt_semantic_seg, train_cfg):
        return self.losses(self.forward(inputs), gt_semantic_seg)

    def forward_test(self, inputs, img_metas, test_cfg):
        return self.forward(inputs)

    def cls_seg(self, feat):
        if self.dropout is not None:
            feat = self.dropout(feat)
        return self.conv_seg(feat)

    def _fused_losses(self, seg_logit, seg_label):
        """MI355X path (csrc/ce_up.hip): bilinear resize + softmax CE + top-1 accuracy in one pass over the
        low-resolution logits; the [B,C,H,W] up-sampled tensor is never materialised.  Used when it computes
        exactly what the generic path below computes (softmax CE, no class weights, align_corners=False)."""
        from .. import ce as hip_ce
        from ..losses import CrossEntropyLoss
        crit = self.loss_decode
        if not (isinstance(crit, CrossEntropyLoss) and crit.class_weight is None and not self.align_corners and self.ignore_index is not None
                and hip_ce.supported(seg_logit, seg_label.shape[2:])):
            return None
        loss_pix, hits = hip_ce.fused_ce_up(seg_logit, seg_label, self.ignore_index)
        if crit.reduction == 'mean':
            loss = loss_pix.mean()
        elif crit.reduction == 'sum':
            loss = loss_pix.sum()
        else:
            loss = loss_pix
        acc = hits * (100.0 / seg_label.numel())          # int32 count times a Python float: ONE kernel, fp32 result
        # loss_weight == 1 (every shipped config): no multiply -- with reduction='none' it was a pass over the [B, H, W] map each way, and its backward
        # turned the broadcast gradient of the map's mean into a dense map the fused CE backward then had to read per class group
        return {'loss_seg': loss if crit.loss_weight == 1.0 else crit.loss_weight * loss, 'acc_seg': acc}

    def losses(self, seg_logit, seg_label):
        fused = self._fused_losses(seg_logit, seg_label)
        if fused is not None:
            return fused
        seg_logit = resize(seg_logit, size=seg_label.shape[2:], mode='bilinear', align_corners=self.align_corners)
        seg_label = seg_label.squeeze(1)
        return {
            'loss_seg': self.loss_decode(seg_logit, seg_label, weight=None, ignore_index=self.ignore_index),
            'acc_seg': accuracy(seg_logit, seg_label),
        }

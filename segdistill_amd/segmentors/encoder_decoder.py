"""backbone -> (neck) -> decode head (+ auxiliary heads) segmentor, train path.

Counterpart of reference mmseg/models/segmentors/encoder_decoder.py (EncoderDecoder :12-166:
extract_feat :77-82, encode_decode :84-94, forward_train :136-166); loss keys are prefixed
``decode.`` / ``aux.`` / ``aux_{i}.`` exactly as there.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import builder
from ..builder import SEGMENTORS
from ..layers import add_prefix, resize
from .base import BaseSegmentor


@SEGMENTORS.register_module()
class EncoderDecoder(BaseSegmentor):
    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        self.decode_head = builder.build_head(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        if auxiliary_head is not None:
            if isinstance(auxiliary_head, list):
                self.auxiliary_head = nn.ModuleList(builder.build_head(c) for c in auxiliary_head)
            else:
                self.auxiliary_head = builder.build_head(auxiliary_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_weights(pretrained=pretrained)
        assert self.with_decode_head

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        self.decode_head.init_weights()
        if self.with_auxiliary_head:
            heads = self.auxiliary_head if isinstance(self.auxiliary_head, nn.ModuleList) else [self.auxiliary_head]
            for h in heads:
                h.init_weights()

    def extract_feat(self, img):
        graphed = getattr(self, '_graphed_backbone', None)   # set by KDTrainer.enable_hybrid_graph
        if graphed is not None and self.training and torch.is_grad_enabled() and img.is_cuda:
            x = graphed(img)
        else:
            x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    def encode_decode(self, img, img_metas=None):
        logits = self.decode_head.forward_test(self.extract_feat(img), img_metas, self.test_cfg)
        return resize(logits, size=img.shape[2:], mode='bilinear', align_corners=self.align_corners)

    def forward_features_only(self, img, run_aux=False):
        """Run backbone + decode head (+ aux heads) WITHOUT computing any loss -- what a frozen
        teacher needs so that its tapped layers fire (the reference instead runs the whole
        forward_train and discards the teacher's CE/accuracy at 512^2)."""
        x = self.extract_feat(img)
        out = self.decode_head(x)
        if run_aux and self.with_auxiliary_head:
            heads = self.auxiliary_head if isinstance(self.auxiliary_head, nn.ModuleList) else [self.auxiliary_head]
            for h in heads:
                h(x)
        return out

    def forward_train(self, img, img_metas, gt_semantic_seg):
        x = self.extract_feat(img)
        losses = add_prefix(self.decode_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg), 'decode')
        if self.with_auxiliary_head:
            if isinstance(self.auxiliary_head, nn.ModuleList):
                for i, h in enumerate(self.auxiliary_head):
                    losses.update(add_prefix(h.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg), f'aux_{i}'))
            else:
                losses.update(add_prefix(self.auxiliary_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg), 'aux'))
        return losses

    def forward_test(self, imgs, img_metas=None, **kwargs):
        """Whole-image inference logits (enough for sanity checks; slide/aug are out of scope)."""
        img = imgs[0] if isinstance(imgs, (list, tuple)) else imgs
        return self.encode_decode(img, img_metas)

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

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..builder import HEADS, build_loss
from ..layers import ConvModule, frozen_derived, resize, tokens_of
from ..linear import call_linear, linear_forward, linear_to_planes, linear_to_planes_supported, token_linear
from .decode_head import BaseDecodeHead


class MLP(nn.Module):
    """[B,C,h,w] -> tokens [B,h*w,E] through one Linear."""

    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)

    def forward(self, x):
        return call_linear(self.proj, tokens_of(x))          # split-K weight gradient at 16384+ tokens per image (linear.py)


# A/B: 0 = the training head runs linear_c_i and its linear_fuse block as two GEMMs per branch (rounds 1-5)
_FOLD_TRAIN = os.environ.get('SEGDISTILL_HEAD_FOLD_TRAIN', '1') == '1'


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

    # ---- forward -------------------------------------------------------------------------------------------------
    # The reference (segformer_head.py:75-98) up-samples the four E-channel maps to 1/4 resolution, concatenates them
    # into a [B, 4E, H/4, W/4] tensor (1.6 GB at B=8, E=768) and applies the 1x1 ``linear_fuse`` conv to it.  A 1x1 conv
    # is linear per pixel and bilinear interpolation is linear over space with weights summing to 1, so they commute:
    #     W . cat_i(up(c_i)) = sum_i up(W_i . c_i),      W_i = W[:, i*E:(i+1)*E]
    # i.e. the fuse conv can run at each branch's NATIVE resolution (64x/16x/4x/1x fewer pixels) and only E-channel
    # results are up-sampled and summed: 3x fewer flops, no concat tensor.  When the head is in eval mode and nobody
    # taps linear_c1..4, the branch Linear is folded into W_i as well (W_i P_i: Cin_i -> E), 25x fewer flops
    # (the frozen B2..B5 teachers).  Same parameters, same state-dict keys, results equal up to fp32 rounding.
    def _branches(self, feats):
        return ((feats[3], self.linear_c4), (feats[2], self.linear_c3), (feats[1], self.linear_c2), (feats[0], self.linear_c1))

    def _branch_maps(self, feats):
        """-> (zs, sizes, n, size, e): the fuse conv applied per branch at native resolution, token-major [B, h_i*w_i, E], finest (c1) first."""
        c1 = feats[0]
        n, size = c1.shape[0], c1.shape[2:]
        w = self.linear_fuse.conv.weight  # [E, 4E, 1, 1], input channel blocks ordered (c4, c3, c2, c1)
        e = w.shape[0]
        hooked = any(m._forward_hooks or m.proj._forward_hooks for _, m in self._branches(feats))
        fold = (not self.training) and not hooked
        # round 6: the same composition while TRAINING (nobody taps linear_c1..4: configs 2 / 3): W_i P_i is a 256 x 256 x Cin product made anew
        # every step INSIDE the autograd graph -- its gradient flows to both factors through two products of the same size -- and the branch is
        # ONE token GEMM Cin -> E each way instead of two (forward, input gradient, weight gradient): the four E x E fuse products over 131072 /
        # 32768 / 8192 / 2048 tokens (68 GF per step with their gradients) disappear.  Equal to the two-GEMM form up to fp32 rounding.
        fold_train = (_FOLD_TRAIN and self.training and torch.is_grad_enabled() and not hooked and c1.is_cuda and w.requires_grad
                      and not torch.is_autocast_enabled())
        zs, sizes = [], []
        # the four [E, E] input-channel blocks as ONE unbind of a strided view: no copy forward, and the backward stacks the four block
        # gradients with one kernel (per-block slicing made autograd zero-fill and add four full-size [E, 4E] gradients)
        w_blocks = w.reshape(e, 4, e).transpose(0, 1).unbind(0) if w.shape[1] == 4 * e else None
        for i, (feat, mlp) in enumerate(self._branches(feats)):
            wi = w_blocks[i] if w_blocks is not None else w[:, i * e:(i + 1) * e, 0, 0]   # [E, E], rows strided by 4E
            tokens = tokens_of(feat)                                  # [B, hw, Cin] (a view for channels-last features)
            if fold:
                # W_i P_i and W_i b_i: four small products per call -- cached while both parameters are frozen (the teacher)
                wp, bp = mlp.proj.weight, mlp.proj.bias
                w_fold = frozen_derived(w, ('fold_w', i), lambda: (wi @ wp).contiguous(), wp)          # [E, Cin]: the layout of a Linear weight
                b_fold = frozen_derived(w, ('fold_b', i), lambda: wi @ bp, bp)
                z = linear_forward(tokens.reshape(-1, tokens.shape[-1]), w_fold, b_fold)              # measured dispatch: library / MFMA kernels
            elif fold_train:
                wp, bp = mlp.proj.weight, mlp.proj.bias
                z = token_linear(tokens.reshape(-1, tokens.shape[-1]), wi @ wp, None if bp is None else wi @ bp, defer_ok=False)
            else:
                z = token_linear(mlp(feat).reshape(-1, e), wi)        # module call keeps forward hooks (taps) alive
            zs.append(z.reshape(n, -1, e))                            # token-major [B, h_i*w_i, E]
            sizes.append(tuple(feat.shape[2:]))
        return zs[::-1], sizes[::-1], n, size, e

    def _fused_sum(self, feats, fold_norm=False):
        """-> (y, normed): the summed branch maps, and whether the eval-mode norm + ReLU were already applied in the same pass."""
        from .. import headfuse
        zs, sizes, n, size, e = self._branch_maps(feats)
        return self._sum_of(zs, sizes, n, size, e, fold_norm)

    def _sum_of(self, zs, sizes, n, size, e, fold_norm):
        from .. import headfuse
        bias = self.linear_fuse.conv.bias
        if fold_norm and headfuse.supported(zs, sizes):
            # frozen network: sum + eval-mode BatchNorm (an affine map per channel) + ReLU in ONE pass; finish() is told to skip them
            norm = self.linear_fuse.norm
            scale = frozen_derived(norm.weight, 'bn_scale', lambda: (norm.weight * torch.rsqrt(norm.running_var + norm.eps)).float(),
                                   norm.running_var)
            shift = frozen_derived(norm.bias, 'bn_shift', lambda: (norm.bias - norm.running_mean * scale).float(), norm.running_mean,
                                   norm.running_var, norm.weight)
            y = headfuse.upsum_affine_inference(zs, bias, sizes, scale, shift, relu=True)
            return y.reshape(n, size[0], size[1], e).permute(0, 3, 1, 2), True
        if headfuse.supported(zs, sizes):
            # MI355X path (csrc/headfuse.hip): one pass, y = z1 + up(z2) + up(z3) + up(z4) + bias, token-major
            y = headfuse.upsum(zs[0], zs[1], zs[2], zs[3], bias, sizes)
            return y.reshape(n, size[0], size[1], e).permute(0, 3, 1, 2), False   # NCHW view with channels-last strides
        total = None
        for z, (h_, w_) in zip(zs, sizes):
            z = z.reshape(n, h_, w_, e).permute(0, 3, 1, 2)
            if (h_, w_) != tuple(size):
                z = resize(z, size=size, mode='bilinear', align_corners=False)
            total = z if total is None else total + z
        if bias is not None:
            total = total + bias.view(1, -1, 1, 1)
        return total, False

    def forward(self, inputs):
        feats = self._transform_inputs(inputs)
        fuse = self.linear_fuse
        if fuse.conv._forward_hooks:
            # someone taps the raw conv output of the concatenation: keep the literal reference dataflow
            c1 = feats[0]
            maps = []
            for feat, proj in self._branches(feats):
                m = proj(feat).permute(0, 2, 1).reshape(c1.shape[0], -1, feat.shape[2], feat.shape[3])
                maps.append(m if m.shape[2:] == c1.shape[2:] else resize(m, size=c1.shape[2:], mode='bilinear', align_corners=False))
            fused = fuse(torch.cat(maps, dim=1))
            if self.dropout is not None:
                fused = self.dropout(fused)
            return self._predict(fused)
        if self._can_fold_norm() and self._can_fuse_pred():
            # frozen network (round 6): sum + norm + ReLU + linear_pred in ONE kernel -- the summed [B, HW, E] map is never written
            from .. import headfuse
            zs, sizes, n, size, e = self._branch_maps(feats)
            pred = self.linear_pred
            if headfuse.head_tail_supported(zs, sizes, pred.out_channels):
                norm = fuse.norm
                scale = frozen_derived(norm.weight, 'bn_scale', lambda: (norm.weight * torch.rsqrt(norm.running_var + norm.eps)).float(),
                                       norm.running_var)
                shift = frozen_derived(norm.bias, 'bn_shift', lambda: (norm.bias - norm.running_mean * scale).float(), norm.running_mean,
                                       norm.running_var, norm.weight)
                out = headfuse.head_tail(zs, sizes, fuse.conv.bias, scale, shift, pred.weight.view(pred.out_channels, e), pred.bias)
                for hook in pred._forward_hooks.values():       # THE tap of every shipped KD config: fired by hand (its input no longer exists)
                    r = hook(pred, (None,), out)
                    if r is not None:
                        out = r
                return out
            y, normed = self._sum_of(zs, sizes, n, size, e, True)
            return self.finish(y, normed)
        y, normed = self._fused_sum(feats, fold_norm=self._can_fold_norm())
        return self.finish(y, normed)

    def _can_fuse_pred(self):
        """linear_pred can join the frozen head's fused pass: a 1x1 conv on a leaf weight nobody differentiates, no dropout in effect, nobody
        observing the fused map (a tap on linear_fuse / dropout) or linear_pred's input."""
        pred, fuse, drop = self.linear_pred, self.linear_fuse, self.dropout
        return (isinstance(pred, nn.Conv2d) and pred.kernel_size == (1, 1) and pred.groups == 1 and not pred.weight.requires_grad
                and pred.weight.is_contiguous() and pred.weight.dtype == torch.float32
                and not (pred._forward_pre_hooks or fuse._forward_hooks or fuse._forward_pre_hooks)
                and (drop is None or not (drop.training and drop.p > 0) and not (drop._forward_hooks or drop._forward_pre_hooks)))

    def _can_fold_norm(self):
        """Eval-mode BatchNorm with running statistics + ReLU, no gradient wanted, nobody hooking either module."""
        fuse = self.linear_fuse
        norm = getattr(fuse, 'norm', None) if fuse.with_norm else None
        return (not torch.is_grad_enabled() and isinstance(norm, nn.modules.batchnorm._BatchNorm) and not norm.training
                and norm.track_running_stats and norm.running_mean is not None and norm.affine and fuse.with_activation
                and isinstance(fuse.activate, nn.ReLU) and not (norm._forward_hooks or norm._forward_pre_hooks or fuse.activate._forward_hooks
                                                                 or fuse.activate._forward_pre_hooks))

    def _fused_tail(self, y):
        """Training-mode (Sync)BatchNorm -> ReLU -> Dropout2d of `linear_fuse` as the four HIP passes of csrc/batchnorm.hip
        (segdistill_amd/batchnorm.py) on the token-major view of y; None when the generic modules must run (hooks, eval mode,
        unsupported layout)."""
        from .. import batchnorm as hip_bn
        fuse, drop = self.linear_fuse, self.dropout
        if not (self.training and fuse.with_norm and fuse.with_activation and isinstance(fuse.activate, nn.ReLU) and y.is_cuda and y.dim() == 4
                and y.is_contiguous(memory_format=torch.channels_last)):
            return None
        mods = (fuse, fuse.activate) + ((drop,) if drop is not None else ())
        if any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in mods):
            return None
        if drop is not None and not (isinstance(drop, nn.Dropout2d) and 0.0 <= drop.p < 1.0):
            return None
        b, e, h, w = y.shape
        tokens = tokens_of(y)
        if not hip_bn.supported(tokens, fuse.norm):
            return None
        scale = hip_bn.channel_dropout_scale(tokens, drop.p) if (drop is not None and drop.training and drop.p > 0.0) else None
        out = hip_bn.norm_act(tokens, fuse.norm, relu=True, drop=scale)
        return out.reshape(b, h, w, e).permute(0, 3, 1, 2)

    def finish(self, y, normed=False):
        """SyncBN -> ReLU -> dropout -> linear_pred on the summed branch maps (the part of the head that may communicate)."""
        fuse = self.linear_fuse
        if not normed:
            fast = self._fused_tail(y)
            if fast is not None:
                return self._predict(fast)
        if fuse.with_norm and not normed:
            y = fuse.norm(y)
        if fuse.with_activation and not normed:
            y = fuse.activate(y)
        fused = y
        if fuse._forward_hooks:  # a tap on linear_fuse itself sees the same output tensor
            for hook in fuse._forward_hooks.values():
                r = hook(fuse, (None,), fused)
                if r is not None:
                    fused = r
        if self.dropout is not None:
            fused = self.dropout(fused)
        return self._predict(fused)

    def _predict(self, fused):
        """linear_pred (1x1 conv, segformer_head.py:73,96).  When `fused` is a channels-last view of tokens the conv is a Linear
        over the tokens; its [B, HW, classes] result is transposed once into the contiguous NCHW planes the loss kernels
        read.  Forward hooks on the module (it is THE tap of every shipped KD config) are fired by hand with that tensor."""
        pred = self.linear_pred
        if fused.is_contiguous() or pred._forward_pre_hooks or not fused.is_cuda:
            return pred(fused)
        b, e, h, w = fused.shape
        tokens = tokens_of(fused)                                              # [B, HW, E] view
        w2d = pred.weight.view(pred.out_channels, e)
        # Linear form + one transpose copy (39-79 MB) rather than W x tokens^T: (a) its weight gradient -- a 150 x E product over
        # 131072 tokens -- then runs on the split-K kernel instead of a 20-workgroup library GEMM (0.39 ms at config 2);
        # (b) ROCm 7.0 hipBLASLt's bf16 kernel for the batched W x tokens^T form (E=768, HW=16384) reads out of bounds
        # (reproduced in isolation in round 1).
        frozen_f32 = not torch.is_grad_enabled() and tokens.dtype == torch.float32 and not torch.is_autocast_enabled()
        if linear_to_planes_supported(tokens, w2d, pred.bias) and (not frozen_f32 or (e % 32 == 0 and _lib.get_tunable('align_split_bf16') == 1)):
            # the swapped-role product writes the class planes directly and its backward reads the gradient planes (csrc: sd_linear_nchw_*):
            # neither the 39-79 MB transpose of the logits nor that of their gradient, and no library bf16 batched GEMM.  Training (fp32 or
            # bf16 storage), the frozen network under bf16 storage, and -- since the 160-row tile -- the frozen fp32 teacher as well
            # (E = 768: 243 us against 299 us for the library's batched W x tokens^T; with 128-row tiles it was 307 us)
            out = linear_to_planes(tokens, w2d, pred.bias)
        elif frozen_f32 and pred.bias is not None:
            # frozen network in fp32 without the split-bf16 kernel (E % 32 != 0, or the mode switched off): the library's W x tokens^T also
            # lands in contiguous NCHW planes directly -- no weight gradient to care about, the faulty library kernel is a bf16 one
            out = torch.baddbmm(pred.bias.view(1, -1, 1), w2d.unsqueeze(0).expand(b, -1, -1), tokens.transpose(1, 2))
        else:
            out = token_linear(tokens, w2d, pred.bias, defer_ok=True).transpose(1, 2).contiguous()   # w2d: a view of the leaf weight
        out = out.view(b, pred.out_channels, h, w)
        for hook in pred._forward_hooks.values():
            r = hook(pred, (fused,), out)
            if r is not None:
                out = r
        return out

"""Mix Transformer (SegFormer MiT-B0..B5) encoder on PyTorch-ROCm.

Behavioural counterpart of the reference's mmseg/models/backbones/mix_transformer.py
(MixVisionTransformer :221-373, Attention :63-133, Mlp :20-55, OverlapPatchEmbed :178-218,
variants mit_b0..b5 :392-441).  Attribute names (patch_embedN.proj/norm, blockN.i.norm1/attn.
{q,kv,sr,norm,proj}/norm2/mlp.{fc1,dwconv.dwconv,fc2}, normN) are the reference's, so its
checkpoints load key-for-key and the distillation taps (``backbone.block1.0.attn.ATTN`` ...)
resolve identically.  The pass-through ``Tap`` children (ATTN/Q/K/V/FEA) exist only to be
hooked by name, as in the reference (:57-61).

MI355X notes: the attention core goes through ``F.scaled_dot_product_attention`` unless the
pre-softmax scores are tapped (a hook on ``ATTN``) or attention dropout is active, in which
case the explicit softmax(QK^T*scale)V form is used so the tapped tensor exists.
"""
from __future__ import annotations

import math
import os
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..builder import BACKBONES
from ..layers import DropPath, frozen_derived, nchw_view_of_tokens, tokens_of, trunc_normal_
from ..layernorm import (HipLayerNorm, add_layernorm, add_layernorm_patches, add_layernorm_supported, layernorm_patches, patch_supported)
from ..linear import call_linear, sr_patch_linear


def _mit_init(m):
    """reference mix_transformer.py:33-46 (same rule in every sub-module)."""
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()


# batch slices the deep stages (3, 4) of a FROZEN encoder are cut into (MixVisionTransformer.forward_features); A/B: SEGDISTILL_DEEP_CHUNKS=1
DEEP_CHUNKS = int(os.environ.get('SEGDISTILL_DEEP_CHUNKS', '2'))
_DEEP_CHUNKS_F32 = os.environ.get('SEGDISTILL_DEEP_CHUNKS_F32', '0') == '1'      # A/B: the slices under fp32 storage too
_DEEP_FROM = int(os.environ.get('SEGDISTILL_DEEP_FROM', '2'))          # first sliced stage (measured on config 5: 3 -> 782, 2 -> 790, 1 -> 791, 4 -> 749 imgs/s)

_LN_PATCHES = True      # test hook: False = the SR path gathers its patches with a copy (the general path)


class Tap(nn.Identity):
    """Identity whose only purpose is to be addressable by a dotted module name."""


class _DepthwiseConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, tokens, hw):
        from .. import dwconv as hip_dw
        conv = self.dwconv
        if hip_dw.supported(tokens, conv.weight):   # fp32 or bf16 storage (autocast hands over bf16 tokens), fp32 accumulation
            # MI355X path: token-major depth-wise kernels, no NCHW round trip (csrc/dwconv.hip)
            return hip_dw.dwconv3x3_tokens(tokens, conv.weight, conv.bias, hw[0], hw[1])
        b, n, c = tokens.shape
        y = conv(tokens.transpose(1, 2).reshape(b, c, *hw))
        return y.flatten(2).transpose(1, 2)


class MixFFN(nn.Module):
    def __init__(self, dim, hidden, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = _DepthwiseConv(hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)

    def forward(self, x, hw):
        h = call_linear(self.fc1, x)
        conv = self.dwconv.dwconv
        from .. import dwconv as hip_dw
        if (not torch.is_grad_enabled() and isinstance(self.act, nn.GELU) and getattr(self.act, 'approximate', 'none') == 'none'
                and hip_dw.supported(h, conv.weight)
                and not (self.dwconv._forward_hooks or conv._forward_hooks or self.act._forward_hooks)):
            from .. import mixffn
            if not (self.training and self.drop.p > 0) and mixffn.usable(h, conv, self.fc2, hw):
                return mixffn.tail(h, conv, self.fc2, hw)                                         # frozen: conv + GELU + fc2 in one kernel
            h = hip_dw.dwconv3x3_gelu_tokens_inference(h, conv.weight, conv.bias, hw[0], hw[1])   # frozen-teacher path
        elif (isinstance(self.act, nn.GELU) and getattr(self.act, 'approximate', 'none') == 'none' and hip_dw.supported(h, conv.weight)
              and not (self.dwconv._forward_hooks or conv._forward_hooks or self.act._forward_hooks)):
            h = hip_dw.dwconv3x3_gelu_tokens(h, conv.weight, conv.bias, hw[0], hw[1])             # training: conv + GELU in one pass
        else:
            h = self.act(self.dwconv(h, hw))
        return self.drop(call_linear(self.fc2, self.drop(h)))


class SRAttention(nn.Module):
    """Multi-head attention whose keys/values come from a spatially reduced map (sr_ratio)."""

    def __init__(self, dim, num_heads, qkv_bias, qk_scale, attn_drop, proj_drop, sr_ratio):
        super().__init__()
        if dim % num_heads:
            raise ValueError(f'dim {dim} should be divided by num_heads {num_heads}.')
        self.dim, self.num_heads = dim, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, 2 * dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.ATTN, self.Q, self.K, self.V = Tap(), Tap(), Tap(), Tap()
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = HipLayerNorm(dim)
            # kept in channels-last STORAGE (same shape / values / state-dict entry, like the patch-embed filters): the (ky, kx, cin)-ordered
            # matrix _spatial_reduce multiplies with is then a VIEW of the parameter -- no re-layout copy forward, none for its gradient
            if os.environ.get('SEGDISTILL_CL_WEIGHTS', '1') == '1':
                self.sr.weight.data = self.sr.weight.data.contiguous(memory_format=torch.channels_last)

    def takes_patches(self, x, hw):
        """True when forward() can take the LayerNorm output ALSO in patch order (`patches=`): the producing LayerNorm kernel then writes it
        (csrc/layernorm.hip, patch form) and its backward gathers the patch-order gradient -- no gather copy, no scatter copy + add."""
        conv = getattr(self, 'sr', None)
        return (_LN_PATCHES and self.sr_ratio > 1 and not (conv._forward_hooks or conv._forward_pre_hooks)
                and conv.weight.is_contiguous(memory_format=torch.channels_last) and patch_supported(x, hw, self.sr_ratio))

    def _spatial_reduce(self, x, hw, patches=None):
        """The SR conv has kernel == stride == r, i.e. it is a Linear over non-overlapping r x r patches.  On token-major
        input that is ONE gather of the patches ([B, H/r, W/r, r*r*C]) and a GEMM; the reference's NCHW route costs a
        transpose copy in, an (often poorly supported) strided conv, and a transpose copy out."""
        r, conv = self.sr_ratio, self.sr
        b, n, c = x.shape
        H, W = hw
        if conv._forward_hooks or H % r or W % r:
            return conv(x.transpose(1, 2).reshape(b, c, H, W)).flatten(2).transpose(1, 2)
        cl = conv.weight.is_contiguous(memory_format=torch.channels_last)
        if patches is not None:          # already gathered by the LayerNorm that produced x
            return sr_patch_linear(patches, conv.weight.permute(0, 2, 3, 1).reshape(conv.out_channels, r * r * c), conv.bias, weight_is_view=True)
        patches = x.reshape(b, H // r, r, W // r, r, c).permute(0, 1, 3, 2, 4, 5).reshape(b, (H // r) * (W // r), r * r * c)
        # (ky, kx, cin) order to match the patches; the re-layout is cached for a frozen network
        if cl:
            w2 = conv.weight.permute(0, 2, 3, 1).reshape(conv.out_channels, r * r * c)     # a view: the storage already has this order
            return sr_patch_linear(patches, w2, conv.bias, weight_is_view=True)
        w2 = frozen_derived(conv.weight, 'sr_patch', lambda: conv.weight.permute(0, 2, 3, 1).reshape(conv.out_channels, r * r * c).contiguous())
        return sr_patch_linear(patches, w2, conv.bias)

    def forward(self, x, hw, patches=None):
        from .. import sra as hip_sra
        b, n, c = x.shape
        h, d = self.num_heads, c // self.num_heads
        q_lin = call_linear(self.q, x)
        src = x
        if self.sr_ratio > 1:
            src = self.norm(self._spatial_reduce(x, hw, patches))
        kv_lin = call_linear(self.kv, src)
        observed = any(t._forward_hooks or t._forward_pre_hooks for t in (self.ATTN, self.Q, self.K, self.V))
        dropping = self.training and self.attn_drop.p > 0
        if not observed and not dropping and hip_sra.supported(q_lin, kv_lin, h) and hip_sra.preferred(n, kv_lin.shape[1], d, q_lin):
            # MI355X path (csrc/sra_attn.hip): all heads in one launch straight from the Linear outputs -- no head transposes,
            # no [B,heads,N,KV] score tensor
            out = hip_sra.sr_attention(q_lin, kv_lin, h, self.scale)
            return self.proj_drop(call_linear(self.proj, out))
        q = self.Q(q_lin.reshape(b, n, h, d).transpose(1, 2))
        kv = kv_lin.reshape(b, -1, 2, h, d).permute(2, 0, 3, 1, 4)
        k, v = self.K(kv[0]), self.V(kv[1])
        explicit = bool(self.ATTN._forward_hooks) or dropping
        # measured on MI355X (round-1 attention probe, fp32): the fused SDPA kernels win everywhere in the forward, but their
        # backward parallelises over the 256 keys only; with >= 8192 queries the explicit form's fwd+bwd is 1.7-2.2x faster
        explicit = explicit or (n >= 8192 and torch.is_grad_enabled() and x.requires_grad)
        if explicit:
            scores = self.ATTN((q @ k.transpose(-2, -1)) * self.scale)
            out = self.attn_drop(scores.softmax(dim=-1)) @ v
        else:
            out = F.scaled_dot_product_attention(q, k, v, scale=self.scale)
        out = out.transpose(1, 2).reshape(b, n, c)
        return self.proj_drop(call_linear(self.proj, out))


class EncoderBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path, norm_layer, sr_ratio):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = SRAttention(dim, num_heads, qkv_bias, qk_scale, attn_drop, drop, sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = MixFFN(dim, int(dim * mlp_ratio), drop)
        self.FEA = Tap()

    def forward(self, x, hw):
        x = x + self.drop_path(self.attn(self.norm1(x), hw))
        x = x + self.drop_path(self.mlp(self.norm2(x), hw))
        return self.FEA(x)

    def _observed(self):
        """True when somebody hooks this block or a module whose output the fused stage loop never materialises on its own."""
        mods = (self, self.norm1, self.norm2, self.drop_path, self.FEA)
        return any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in mods)

    def _scale(self, x):
        """Stochastic-depth factor [B] of the next residual branch (None = identity): a row of the table the encoder drew for the
        whole forward (MixVisionTransformer._draw_drop_path) when there is one, else an own draw."""
        pending = getattr(self, '_dp_pending', None)
        if pending:
            return pending.pop(0)
        return self.drop_path.sample_scale(x) if isinstance(self.drop_path, DropPath) else None


def _run_stage(x, hw, blocks, stage_norm):
    """blocks + stage norm on tokens x.  MI355X path: every residual add (with its stochastic-depth factor) is fused with the
    LayerNorm that consumes the sum -- the block's norm2, the next block's norm1, or the stage norm -- so the residual stream
    is read and written once per half-block instead of three times (csrc/layernorm.hip, residual form).  Falls back to the
    literal block-by-block form when a hook observes an intermediate or the shape is unsupported."""
    norms = [b.norm1 for b in blocks[1:]] + [stage_norm]
    fused = (len(blocks) > 0 and not any(b._observed() for b in blocks)
             and all(add_layernorm_supported(x, x, n) for b, nx in zip(blocks, norms) for n in (b.norm2, nx)))
    if not fused:
        for blk in blocks:
            x = blk(x, hw)
        return stage_norm(x)
    # a LayerNorm whose output feeds a spatial-reduction attention also writes it in the SR conv's patch order (layernorm.py, patch form)
    want = [blk.attn.takes_patches(x, hw) for blk in blocks]
    pt = None
    if want[0]:
        n, pt = layernorm_patches(x, blocks[0].norm1, hw, blocks[0].attn.sr_ratio)
    else:
        n = blocks[0].norm1(x)
    for i, (blk, nxt) in enumerate(zip(blocks, norms)):
        x, n = add_layernorm(x, blk.attn(n, hw, patches=pt) if pt is not None else blk.attn(n, hw), blk.norm2, blk._scale(x))
        if i + 1 < len(blocks) and want[i + 1]:
            x, n, pt = add_layernorm_patches(x, blk.mlp(n, hw), nxt, hw, blocks[i + 1].attn.sr_ratio, blk._scale(x))
        else:
            x, n = add_layernorm(x, blk.mlp(n, hw), nxt, blk._scale(x))
            pt = None
    return n


class _ConvDeferredBias(torch.autograd.Function):
    """F.conv2d whose bias gradient joins the batched column-sum pass of the backward (segdistill_amd/deferred.py) instead of being one ATen
    reduction per layer inside convolution_backward (4 x 14 us per config-2 step, at ~1.2 TB/s): the incoming gradient of a channels-last
    convolution output IS a token-major [B*H*W, C] matrix.  Input and weight gradients stay the library's.  fp32 storage only: the same under
    bf16 autocast (operand casts from the parameters' shadows, 16 -> 8 cast kernels per step) measured neutral on config 5 and was not kept.
    Config 2, same box: 780.6 - 784.3 imgs/s with it, 774.9 / 778.6 without (profiles/r04_ab_cfg2_conv_deferred_bias.txt)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        ctx.save_for_backward(x, weight)
        ctx.conf = (stride, padding, dilation, groups, bias.dtype)
        return F.conv2d(x, weight, bias, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, dy):
        from .. import deferred
        x, weight = ctx.saved_tensors
        stride, padding, dilation, groups, bdtype = ctx.conf
        dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, stride, padding, dilation, False, [0, 0], groups,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        db = None
        if ctx.needs_input_grad[2]:
            if dy.dim() == 4 and dy.is_contiguous(memory_format=torch.channels_last):
                db = deferred.column_sum(dy.permute(0, 2, 3, 1).reshape(-1, dy.shape[1]), True).to(bdtype)
            else:
                db = dy.sum((0, 2, 3)).to(bdtype)
        return dx, dw, db, None, None, None, None


_CONV_DEFERRED_BIAS = True      # test hook: False = ATen's per-layer reduction (the general path)


class OverlapPatchEmbed(nn.Module):
    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = HipLayerNorm(embed_dim)
        # The stage inputs are channels-last views of tokens, so MIOpen runs its NHWC kernels and ATen would re-lay the filter out
        # on every call (forward and backward).  The parameter itself is kept in channels-last STORAGE from construction on
        # (same shape, same values, same state-dict entry; `.to(device)`, in-place init and checkpoint loads preserve it).
        if os.environ.get('SEGDISTILL_CL_WEIGHTS', '1') == '1':
            self.proj.weight.data = self.proj.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x):
        p = self.proj
        from .. import patch_embed as hip_pe
        if hip_pe.supported(x, p):
            # MI355X path (csrc/patch_embed.hip): window gather + token GEMM -- deterministic filter gradients, the output already token-major
            tokens, hw = hip_pe.patch_embed_tokens(x, p)
            return self.norm(tokens), hw
        if (x.is_cuda and torch.is_autocast_enabled() and p.weight.dtype == torch.float32 and not p.weight.requires_grad
                and (p.bias is None or not p.bias.requires_grad) and not (p._forward_hooks or p._forward_pre_hooks)):
            # frozen network under low-precision storage (the config-5 teacher): autocast would cast filter and bias again on every call --
            # 8 cast kernels per step for the four stages; the copies are cached per parameter (frozen_derived; strides preserved)
            dt = torch.get_autocast_dtype('cuda')
            w = frozen_derived(p.weight, ('cast', dt), lambda: p.weight.to(dt))
            b = None if p.bias is None else frozen_derived(p.bias, ('cast', dt), lambda: p.bias.to(dt))
            x = F.conv2d(x if x.dtype == dt else x.to(dt), w, b, p.stride, p.padding, p.dilation, p.groups)
        elif (_CONV_DEFERRED_BIAS and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and torch.is_grad_enabled()
              and p.bias is not None and p.bias.requires_grad and p.weight.requires_grad and not (p._forward_hooks or p._forward_pre_hooks)):
            x = _ConvDeferredBias.apply(x, p.weight, p.bias, p.stride, p.padding, p.dilation, p.groups)
        else:
            x = p(x)
        hw = tuple(x.shape[2:])
        return self.norm(tokens_of(x)), hw


class MixVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=(64, 128, 256, 512),
                 num_heads=(1, 2, 4, 8), mlp_ratios=(4, 4, 4, 4), qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1)):
        super().__init__()
        self.num_classes = num_classes
        self.depths = tuple(depths)
        self.embed_dims = tuple(embed_dims)
        chans = (in_chans,) + tuple(embed_dims)
        rates = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        cursor = 0
        for s in range(4):
            setattr(self, f'patch_embed{s + 1}', OverlapPatchEmbed(7 if s == 0 else 3, 4 if s == 0 else 2, chans[s], chans[s + 1]))
        for s in range(4):
            blocks = nn.ModuleList(
                EncoderBlock(embed_dims[s], num_heads[s], mlp_ratios[s], qkv_bias, qk_scale, drop_rate, attn_drop_rate,
                             rates[cursor + i], norm_layer, sr_ratios[s]) for i in range(depths[s]))
            cursor += depths[s]
            setattr(self, f'block{s + 1}', blocks)
            setattr(self, f'norm{s + 1}', norm_layer(embed_dims[s]))
        self.apply(_mit_init)

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)

    def reset_drop_path(self, drop_path_rate):
        rates = torch.linspace(0, drop_path_rate, sum(self.depths)).tolist()
        cursor = 0
        for s in range(4):
            for i, blk in enumerate(getattr(self, f'block{s + 1}')):
                if isinstance(blk.drop_path, DropPath):
                    blk.drop_path.drop_prob = rates[cursor + i]
            cursor += self.depths[s]

    def _draw_drop_path(self, x):
        """All stochastic-depth factors of one forward with TWO kernels (one Bernoulli over a [2*blocks, B] table of keep
        probabilities, one division) instead of two per residual branch (28 tiny launches for B0): each block gets two rows, one
        per branch, drawn independently as the reference's two `self.drop_path(...)` calls are (mix_transformer.py:150-151)."""
        blocks = [b for s in range(1, 5) for b in getattr(self, f'block{s}')]
        for b in blocks:
            b._dp_pending = None
        if not (self.training and x.is_cuda):
            return
        active = [b for b in blocks if isinstance(b.drop_path, DropPath) and b.drop_path.drop_prob > 0.]
        if not active:
            return
        probs = tuple(1.0 - b.drop_path.drop_prob for b in active for _ in range(2))
        cache = getattr(self, '_dp_keep', None)
        if cache is None or cache[0] != (probs, x.device):
            cache = ((probs, x.device), torch.tensor(probs, dtype=torch.float32, device=x.device).unsqueeze(1))
            object.__setattr__(self, '_dp_keep', cache)
        keep = cache[1]
        table = torch.bernoulli(keep.expand(-1, x.shape[0])).div_(keep)
        for i, b in enumerate(active):
            b._dp_pending = [table[2 * i], table[2 * i + 1]]

    def _stage(self, s, x):
        x, hw = getattr(self, f'patch_embed{s}')(x)
        x = _run_stage(x, hw, list(getattr(self, f'block{s}')), getattr(self, f'norm{s}'))
        return x, hw

    def _deep_stages_observed(self):
        """True when a hook sits on any module of stages 3-4 (a tap there must see ONE call with the whole batch)."""
        cached = getattr(self, '_deep_mods', None)
        if cached is None:
            cached = [m for s in range(_DEEP_FROM, 5) for top in (getattr(self, f'patch_embed{s}'), getattr(self, f'block{s}'), getattr(self, f'norm{s}'))
                      for m in top.modules()]
            object.__setattr__(self, '_deep_mods', cached)
        return any(m._forward_hooks or m._forward_pre_hooks for m in cached)

    def forward_features(self, x):
        feats = []
        self._draw_drop_path(x)
        # MI355X (round 6): a FROZEN encoder (no graph to build: the teacher) runs its deep stages (2-4) -- 2048-32768 tokens per kernel, 5-15 us each,
        # most of the 256 CUs idle -- as DEEP_CHUNKS concurrent chains over slices of the batch, each on a stream forked from / joined into the
        # calling one (inside a capture: parallel branches of the hipGraph).  Per-image arithmetic: nothing in these stages mixes images.
        # MEASURED (profiles/r06_ab_deep_chunks.txt, same box): config 5 (bf16 storage, 30 + 2 deep blocks of the B4 teacher) 735 -> 759 imgs/s with
        # two slices; config 2 (fp32, B2 teacher) 792 -> 778: at half the tokens the fp32 Linears fall out of the split-bf16 planes kernels'
        # dispatch (linear._gemm_mode) -- so fp32 storage keeps one chain; four slices lose everywhere (559 / 642).
        n = DEEP_CHUNKS
        split = (n > 1 and x.is_cuda and not torch.is_grad_enabled() and not self.training and x.shape[0] % n == 0 and x.shape[0] >= 2 * n
                 and (torch.is_autocast_enabled() or _DEEP_CHUNKS_F32) and not self._deep_stages_observed())
        for s in range(1, 5):
            if split and s == _DEEP_FROM:
                break
            x, hw = self._stage(s, x)
            # logically [B,C,H,W] like the reference (:340,:347,...), but as a channels-last VIEW of the tokens on the GPU:
            # the next stage's conv and the head consume it without the two transpose copies per stage
            x = nchw_view_of_tokens(x, hw) if x.is_cuda else x.reshape(x.shape[0], hw[0], hw[1], -1).permute(0, 3, 1, 2).contiguous()
            feats.append(x)
        if not split:
            return feats
        cur = torch.cuda.current_stream(x.device)
        streams = getattr(self, '_chunk_streams', None)
        if streams is None or len(streams) < n - 1 or streams[0].device != x.device:
            streams = [torch.cuda.Stream(device=x.device) for _ in range(n - 1)]
            object.__setattr__(self, '_chunk_streams', streams)
        fork = torch.cuda.Event()
        fork.record(cur)
        outs, joins = [], []
        for i, piece in enumerate(x.chunk(n)):
            st = cur if i == 0 else streams[i - 1]
            if i:
                st.wait_event(fork)
            with torch.cuda.stream(st):
                toks = []
                y = piece
                for s in range(_DEEP_FROM, 5):
                    t, hw = self._stage(s, y)
                    toks.append((t, hw))
                    y = nchw_view_of_tokens(t, hw)
                outs.append(toks)
                if i:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    joins.append(ev)
        for ev in joins:
            cur.wait_event(ev)
        for k in range(5 - _DEEP_FROM):
            for o in outs[1:]:
                o[k][0].record_stream(cur)            # allocated on a slice's stream, read by the concatenation on the calling one
            feats.append(nchw_view_of_tokens(torch.cat([o[k][0] for o in outs], 0), outs[0][k][1]))
        return feats

    def forward(self, x):
        return self.forward_features(x)


_VARIANTS = {  # embed_dims, depths  (reference mix_transformer.py:392-441)
    'mit_b0': ((32, 64, 160, 256), (2, 2, 2, 2)),
    'mit_b1': ((64, 128, 320, 512), (2, 2, 2, 2)),
    'mit_b2': ((64, 128, 320, 512), (3, 4, 6, 3)),
    'mit_b3': ((64, 128, 320, 512), (3, 4, 18, 3)),
    'mit_b4': ((64, 128, 320, 512), (3, 8, 27, 3)),
    'mit_b5': ((64, 128, 320, 512), (3, 6, 40, 3)),
}


def _make_variant(name, dims, depths):
    def __init__(self, **kwargs):  # extra config keys such as style='pytorch' are accepted and ignored, as in the reference
        MixVisionTransformer.__init__(self, patch_size=4, embed_dims=dims, num_heads=(1, 2, 5, 8), mlp_ratios=(4, 4, 4, 4),
                                      qkv_bias=True, norm_layer=partial(HipLayerNorm, eps=1e-6), depths=depths,
                                      sr_ratios=(8, 4, 2, 1), drop_rate=0.0, drop_path_rate=0.1)
    cls = type(name, (MixVisionTransformer,), {'__init__': __init__, '__doc__': f'SegFormer {name} encoder.'})
    return BACKBONES.register_module()(cls)


for _n, (_d, _p) in _VARIANTS.items():
    globals()[_n] = _make_variant(_n, _d, _p)

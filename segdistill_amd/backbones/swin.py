"""Swin Transformer encoder (frozen teacher of BASELINE config 4) on PyTorch-ROCm.

Behavioural counterpart of reference mmseg/models/backbones/swin_transformer.py
(WindowAttention :72-151, SwinTransformerBlock :154-260, PatchMerging :263-292, BasicLayer
:295-393, PatchEmbed :396-437, SwinTransformer :440-618).  Parameter / buffer names match the
reference so its checkpoints load key-for-key:
``patch_embed.{proj,norm}``, ``layers.i.blocks.j.{norm1, attn.{relative_position_bias_table,
relative_position_index, qkv, proj}, norm2, mlp.{fc1,fc2}}``, ``layers.i.downsample.{reduction,norm}``,
``norm{i}``.

MI355X notes: the shifted-window mask is built once per (padded size, device) and cached
instead of being rebuilt by a Python double loop on every forward of every stage; the window
attention adds (relative-position bias + mask) as one additive term of SDPA.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..builder import BACKBONES
from ..layernorm import HipLayerNorm, layernorm_map, map_supported          # nn.LayerNorm subclass: same keys, HIP kernels on CUDA tokens (ATen's ran 3.9 ms of a config-4 step)
from .. import window_attn
from ..layers import DropPath, to_2tuple, trunc_normal_


_GATHER_WINDOWS = True          # test hook: False = the literal pad / roll / window_partition chain (the path a trainable Swin takes)
_FUSE_NORMS = os.environ.get('SEGDISTILL_SWIN_FUSE_NORMS', '1') == '1'      # A/B: 0 = LayerNorm, zero-row cat, index_select and add as separate kernels
_TABLES = {}


def _window_tables(h, w, ws, shift, device):
    """(fwd, inv) int64 index tables of one image: fwd[j] = token that lands at window-ordered position j of the padded, cyclically shifted
    map (h*w = the appended zero row for padded positions); inv[t] = window-ordered position of token t.  Built from the module's own
    pad / roll / window_partition on an index map, so the gather IS that composition."""
    key = (h, w, ws, shift, str(device))
    hit = _TABLES.get(key)
    if hit is None:
        pad_r, pad_b = (ws - w % ws) % ws, (ws - h % ws) % ws
        idx = torch.arange(h * w).view(1, h, w, 1)
        idx = F.pad(idx, (0, 0, 0, pad_r, 0, pad_b), value=h * w)
        if shift > 0:
            idx = torch.roll(idx, shifts=(-shift, -shift), dims=(1, 2))
        fwd = window_partition(idx, ws).reshape(-1)
        inv = torch.empty(h * w + 1, dtype=torch.long)
        inv[fwd] = torch.arange(fwd.numel())
        fwd, inv = fwd.to(device), inv[:h * w].contiguous().to(device)
        hit = (fwd, inv, fwd.int(), inv.int())          # int32 copies: the row maps of layernorm_map
        _TABLES[key] = hit
    return hit


def window_partition(x, ws):
    """[B,H,W,C] -> [B*nW, ws, ws, C]"""
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)


def window_reverse(windows, ws, h, w):
    b = windows.shape[0] // ((h // ws) * (w // ws))
    x = windows.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


class Mlp(nn.Module):
    def __init__(self, dim, hidden, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        wh, ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        ys, xs = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing='ij')
        coords = torch.stack([ys.flatten(), xs.flatten()])                 # [2, N]
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0)   # [N, N, 2]
        index = (rel[..., 0] + wh - 1) * (2 * ww - 1) + (rel[..., 1] + ww - 1)
        self.register_buffer('relative_position_index', index)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)

    def _bias_packed(self, n, h):
        """relative position bias [h, N, N] in the attention kernel's accumulator layout (padded keys = -inf); cached for a frozen table."""
        tbl = self.relative_position_bias_table
        sig = (tbl._version, tbl.data_ptr(), n)
        hit = None if tbl.requires_grad else getattr(self, '_bias_p_cache', None)     # a trainable table is gathered every call: fused
        if hit is not None and hit[0] == sig:                                          # optimizers do not bump _version
            return hit[1]
        with torch.no_grad():
            bp = window_attn.pack_tables(tbl[self.relative_position_index.view(-1)].view(n, n, h).permute(2, 0, 1), float('-inf'))[0]
        if not tbl.requires_grad and not (tbl.is_cuda and torch.cuda.is_current_stream_capturing()):
            object.__setattr__(self, '_bias_p_cache', (sig, bp))
        return bp

    def _mask_packed(self, mask):
        if mask is None:
            return None, None
        sig = (mask.data_ptr(), mask._version, tuple(mask.shape))
        hit = getattr(self, '_mask_p_cache', None)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            mp = window_attn.pack_tables(mask, 0.0, True)
        if not (mask.is_cuda and torch.cuda.is_current_stream_capturing()):
            object.__setattr__(self, '_mask_p_cache', (sig, mp))
        return mp

    def forward(self, x, mask=None):
        """x: [nW*B, N, C]; mask: [nW, N, N] additive (0 / -100) or None."""
        bw, n, c = x.shape
        h = self.num_heads
        tbl = self.relative_position_bias_table
        qkv = self.qkv(x)
        if (not (torch.is_grad_enabled() and (qkv.requires_grad or tbl.requires_grad)) and not (self.training and self.attn_drop.p > 0)
                and window_attn.supported(qkv, n, h, c // h) and (mask is None or bw % mask.shape[0] == 0)):
            # no graph to build (the frozen teacher): one kernel over the qkv Linear's output, bias and mask read from their tables
            # (csrc/window_attn.hip) -- no [windows, heads, N, N] additive tensor, no permuted copies of q / k / v or of the output
            out = window_attn.forward_packed(qkv.contiguous(), self._bias_packed(n, h), *self._mask_packed(mask), h, self.scale)
            return self.proj_drop(self.proj(out))
        q, k, v = qkv.reshape(bw, n, 3, h, c // h).permute(2, 0, 3, 1, 4)
        def additive():
            bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, h).permute(2, 0, 1)  # [h,N,N]
            add = bias.unsqueeze(0)
            if mask is not None:
                nw = mask.shape[0]
                add = (add + mask.unsqueeze(1)).repeat(bw // nw, 1, 1, 1)  # windows of one image are contiguous
            return add.to(q.dtype)
        if tbl.requires_grad:
            add = additive()
        else:
            # frozen network (the config-4 teacher): bias gather + mask add + the repeat over the batch -- a 111 MB copy per stage-1 block --
            # are the same tensor every step.  ONE entry per module, the most recent geometry (keyed entries were never evicted, so
            # an eval run over variable-size images grew by one [B*nW, h, N, N] tensor per distinct size and shifted block); a checkpoint load
            # bumps the table's version and invalidates it; nothing is stored while a hipGraph is being captured
            sig = (tbl._version, tbl.data_ptr(), None if mask is None else (mask.data_ptr(), mask._version), bw, q.dtype)
            hit = getattr(self, '_add_cache', None)
            if hit is not None and hit[0] == sig:
                add = hit[1]
            else:
                with torch.no_grad():
                    add = additive()
                if not (q.is_cuda and torch.cuda.is_current_stream_capturing()):
                    object.__setattr__(self, '_add_cache', (sig, add))
        drop = self.attn_drop.p if self.training else 0.
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=add, dropout_p=drop, scale=self.scale)
        return self.proj_drop(self.proj(out.transpose(1, 2).reshape(bw, n, c)))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=HipLayerNorm):
        super().__init__()
        assert 0 <= shift_size < window_size, 'shift_size must in 0-window_size'
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, to_2tuple(window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop)
        self.H = self.W = None

    def fusable(self, x):
        """no graph to build, nothing stochastic, both norms on the HIP kernels: forward_fused applies"""
        return (_GATHER_WINDOWS and _FUSE_NORMS and (not self.training or isinstance(self.drop_path, nn.Identity))
                and map_supported(x, self.norm1) and map_supported(x, self.norm2)
                # forward_fused runs forward-only kernels around the attention / MLP modules: with ANY trainable parameter in the block (frozen
                # norms, trainable qkv / proj / bias table / MLP) their outputs would enter those kernels and the gradients would be lost
                and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
                # a tapped block (Extractor hooks) must be CALLED and must return its complete output: forward_fused does neither
                and not (self._forward_hooks or self._forward_pre_hooks or self.drop_path._forward_hooks or self.drop_path._forward_pre_hooks))

    def forward_fused(self, x, mask_matrix, pending=None):
        """The block of a frozen network as  norm1 + pad + shift + window partition  (one kernel, folding in the previous block's pending MLP
        residual), qkv / attention / proj,  window reverse + un-shift + un-pad + residual add + norm2  (one kernel), MLP.  Returns
        (x after the attention residual, MLP output): the caller adds the second to the first -- the next block does it inside its first
        kernel.  Replaces per block: LayerNorm, zero-row cat, two index_selects and two adds."""
        b, n, c = x.shape
        h, w, ws = self.H, self.W, self.window_size
        assert n == h * w, 'input feature has wrong size'
        _, _, fwd32, inv32 = _window_tables(h, w, ws, self.shift_size, x.device)
        xs, win = layernorm_map(x, self.norm1, res=pending, x_map=fwd32)
        if xs is None:
            xs = x
        att = self.attn(win.view(-1, ws * ws, c), mask=mask_matrix if self.shift_size > 0 else None)
        x1, y2 = layernorm_map(xs, self.norm2, res=att.view(b, -1, c), res_map=inv32)
        return x1, self.mlp(y2)

    def forward(self, x, mask_matrix):
        b, n, c = x.shape
        h, w = self.H, self.W
        assert n == h * w, 'input feature has wrong size'
        ws = self.window_size
        pad_r, pad_b = (ws - w % ws) % ws, (ws - h % ws) % ws
        if x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad) and _GATHER_WINDOWS:
            # no autograd graph to build (the frozen teacher): pad + cyclic shift + window partition are ONE row gather (they were three
            # copies), window reverse + un-shift + un-pad another (index tables cached per geometry; padded positions read an appended
            # zero row, exactly what F.pad produced)
            fwd_idx, inv_idx = _window_tables(h, w, ws, self.shift_size, x.device)[:2]
            y = self.norm1(x)
            y = torch.cat([y, y.new_zeros(b, 1, c)], 1) if (pad_r or pad_b) else y
            win = y.index_select(1, fwd_idx).view(-1, ws * ws, c)
            win = self.attn(win, mask=mask_matrix if self.shift_size > 0 else None)
            y = win.reshape(b, -1, c).index_select(1, inv_idx)
            x = x + self.drop_path(y)
            return x + self.drop_path(self.mlp(self.norm2(x)))
        y = self.norm1(x).view(b, h, w, c)
        if pad_r or pad_b:
            y = F.pad(y, (0, 0, 0, pad_r, 0, pad_b))
        hp, wp = h + pad_b, w + pad_r
        mask = None
        if self.shift_size > 0:
            y = torch.roll(y, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
            mask = mask_matrix
        win = window_partition(y, ws).view(-1, ws * ws, c)
        win = self.attn(win, mask=mask).view(-1, ws, ws, c)
        y = window_reverse(win, ws, hp, wp)
        if self.shift_size > 0:
            y = torch.roll(y, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        if pad_r or pad_b:
            y = y[:, :h, :w, :].contiguous()
        x = x + self.drop_path(y.reshape(b, h * w, c))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=HipLayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x, h, w):
        b, n, c = x.shape
        assert n == h * w, 'input feature has wrong size'
        x = x.view(b, h, w, c)
        if h % 2 or w % 2:
            x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2))
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], dim=-1)
        return self.reduction(self.norm(x.view(b, -1, 4 * c)))


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., norm_layer=HipLayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        if use_checkpoint:
            raise NotImplementedError('activation checkpointing is outside the KD path (the teacher runs under no_grad)')
        self.window_size, self.shift_size, self.depth = window_size, window_size // 2, depth
        self.blocks = nn.ModuleList(
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias, qk_scale,
                                 drop, attn_drop, drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer)
            for i in range(depth))
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None
        self._mask_cache = {}

    def _shift_mask(self, hp, wp, device):
        key = (hp, wp, str(device))
        m = self._mask_cache.get(key)
        if m is None:
            ws, sh = self.window_size, self.shift_size
            region = torch.zeros((1, hp, wp, 1), device=device)
            spans = (slice(0, -ws), slice(-ws, -sh), slice(-sh, None))
            for i, hs in enumerate(spans):
                for j, wsl in enumerate(spans):
                    region[:, hs, wsl, :] = i * 3 + j
            flat = window_partition(region, ws).view(-1, ws * ws)
            diff = flat.unsqueeze(1) - flat.unsqueeze(2)
            m = torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))
            self._mask_cache[key] = m
        return m

    def forward(self, x, h, w):
        ws = self.window_size
        hp, wp = -(-h // ws) * ws, -(-w // ws) * ws
        mask = self._shift_mask(hp, wp, x.device)
        pend = None                                    # MLP output of the previous fused block, not yet added to x
        for blk in self.blocks:
            blk.H, blk.W = h, w
            if blk.fusable(x):
                x, pend = blk.forward_fused(x, mask, pend)
                continue
            if pend is not None:
                x, pend = x + pend, None
            x = blk(x, mask)
        if pend is not None:
            x = x + pend
        if self.downsample is not None:
            return x, h, w, self.downsample(x, h, w), (h + 1) // 2, (w + 1) // 2
        return x, h, w, x, h, w


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size = to_2tuple(patch_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        ph, pw = self.patch_size
        h, w = x.shape[2:]
        if w % pw or h % ph:
            x = F.pad(x, (0, (pw - w % pw) % pw, 0, (ph - h % ph) % ph))
        x = self.proj(x)
        if self.norm is not None:
            hh, ww = x.shape[2:]
            x = self.norm(x.flatten(2).transpose(1, 2)).transpose(1, 2).reshape(-1, self.embed_dim, hh, ww)
        return x


@BACKBONES.register_module()
class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2,
                 norm_layer=HipLayerNorm, ape=False, patch_norm=True, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                 use_checkpoint=False):
        super().__init__()
        self.num_layers = len(depths)
        self.embed_dim, self.ape, self.patch_norm = embed_dim, ape, patch_norm
        self.out_indices, self.frozen_stages = tuple(out_indices), frozen_stages
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, norm_layer if patch_norm else None)
        if ape:
            pi, ps = to_2tuple(pretrain_img_size), to_2tuple(patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, pi[0] // ps[0], pi[1] // ps[1]))
            trunc_normal_(self.absolute_pos_embed, std=.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        rates = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias, qk_scale,
                                          drop_rate, attn_drop_rate, rates[sum(depths[:i]):sum(depths[:i + 1])], norm_layer,
                                          PatchMerging if i < self.num_layers - 1 else None, use_checkpoint))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in self.out_indices:
            self.add_module(f'norm{i}', norm_layer(self.num_features[i]))
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1 and self.ape:
            self.absolute_pos_embed.requires_grad = False
        if self.frozen_stages >= 2:
            self.pos_drop.eval()
            for i in range(self.frozen_stages - 1):
                self.layers[i].eval()
                for p in self.layers[i].parameters():
                    p.requires_grad = False

    def init_weights(self, pretrained=None):
        def _init(m):
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.zeros_(m.bias)
                nn.init.ones_(m.weight)
        if pretrained is not None and not isinstance(pretrained, str):
            raise TypeError('pretrained must be a str or None')
        self.apply(_init)
        if isinstance(pretrained, str):
            from ..checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)

    def forward(self, x):
        x = self.patch_embed(x)
        h, w = x.shape[2:]
        if self.ape:
            x = x + F.interpolate(self.absolute_pos_embed, size=(h, w), mode='bicubic')
        x = self.pos_drop(x.flatten(2).transpose(1, 2))
        outs = []
        for i, layer in enumerate(self.layers):
            y, oh, ow, x, h, w = layer(x, h, w)
            if i in self.out_indices:
                y = getattr(self, f'norm{i}')(y)
                outs.append(y.view(-1, oh, ow, self.num_features[i]).permute(0, 3, 1, 2).contiguous())
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

"""Training-mode (Sync)BatchNorm + ReLU + channel dropout on token-major activations: autograd bindings of
csrc/batchnorm.hip.

Counterpart of the `linear_fuse` tail of the reference's SegFormer head -- SyncBN -> ReLU (segformer_head.py:66-71,
93-94) -> Dropout2d (decode_head.py:210-215) -- as four HBM-bound passes (statistics, apply, backward sums, backward dx)
instead of ATen's batch-norm kernels plus separate ReLU / dropout passes each way.  With more than one rank the two
collectives of torch.nn.SyncBatchNorm sit between the passes: the all-gather of (mean, invstd, count) -- combined by
torch.batch_norm_gather_stats_with_counts, exactly as torch does -- and the all-reduce of the dy sums.  Each collective is
handed to a `cut` callable: a plain call in eager execution, ``SegmentRecorder.cut`` while a segmented hipGraph step is
being recorded (engine/segments.py), where it ends one graph and opens the next.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch.nn.modules.batchnorm import _BatchNorm

from . import _lib
from .ops import _DT, _stream_ptr


def _call(fn):
    fn()


def supported(tokens, norm):
    """tokens [B, N, C] contiguous on the GPU; `norm` a BatchNorm layer in training mode with learnable affine terms."""
    return (isinstance(norm, _BatchNorm) and norm.training and tokens.is_cuda and tokens.dtype in _DT and tokens.dim() == 3
            and tokens.is_contiguous() and tokens.shape[-1] % 4 == 0 and tokens.shape[-1] <= 1024 and tokens.numel() > 0
            and tokens.shape[-1] == norm.num_features and norm.affine and norm.momentum is not None
            and norm.weight.dtype == torch.float32
            and not (norm._forward_hooks or norm._forward_pre_hooks or norm._backward_hooks))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _workspace(rows, C, device):
    n = _lib.lib().sd_bn_workspace_bytes(rows, C)
    return torch.empty(n, dtype=torch.uint8, device=device), n


# ---- the four passes (no autograd) ------------------------------------------------------------------------------------
def local_stats(xt, eps, running_mean=None, running_var=None, momentum=0.0):
    C = xt.shape[-1]
    rows = xt.numel() // C
    mean = torch.empty(C, dtype=torch.float32, device=xt.device)
    invstd = torch.empty(C, dtype=torch.float32, device=xt.device)
    ws, n = _workspace(rows, C, xt.device)
    rc = _lib.lib().sd_bn_stats(xt.data_ptr(), _DT[xt.dtype], rows, C, float(eps), mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean),
                                _ptr(running_var), float(momentum), ws.data_ptr(), n, _stream_ptr())
    _lib.check(rc, 'sd_bn_stats')
    return mean, invstd


def apply_fwd(xt, mean, invstd, w, b, drop, relu):
    C = xt.shape[-1]
    rows = xt.numel() // C
    y = torch.empty_like(xt)
    rc = _lib.lib().sd_bn_act_fwd(xt.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(w), _ptr(b), _ptr(drop), xt.shape[1], int(relu),
                                  y.data_ptr(), _DT[xt.dtype], rows, C, _stream_ptr())
    _lib.check(rc, 'sd_bn_act_fwd')
    return y


def bwd_reduce(xt, dy, mean, invstd, w, b, drop, relu):
    C = xt.shape[-1]
    rows = xt.numel() // C
    sums = torch.empty(2, C, dtype=torch.float32, device=xt.device)   # [sum_dy; sum_dy_xmu]: one buffer, one all-reduce
    ws, n = _workspace(rows, C, xt.device)
    rc = _lib.lib().sd_bn_act_bwd_reduce(xt.data_ptr(), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(w), _ptr(b), _ptr(drop),
                                         xt.shape[1], int(relu), sums[0].data_ptr(), sums[1].data_ptr(), _DT[xt.dtype], rows, C,
                                         ws.data_ptr(), n, _stream_ptr())
    _lib.check(rc, 'sd_bn_act_bwd_reduce')
    return sums


def bwd_elemt(xt, dy, mean, invstd, w, b, drop, relu, sums, inv_count, inv_count_dev=None):
    C = xt.shape[-1]
    rows = xt.numel() // C
    dx = torch.empty_like(xt)
    rc = _lib.lib().sd_bn_act_bwd_elemt(xt.data_ptr(), dy.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(w), _ptr(b), _ptr(drop),
                                        xt.shape[1], int(relu), sums[0].data_ptr(), sums[1].data_ptr(), float(inv_count), _ptr(inv_count_dev),
                                        dx.data_ptr(), _DT[xt.dtype], rows, C, _stream_ptr())
    _lib.check(rc, 'sd_bn_act_bwd_elemt')
    return dx


# ---- forward / backward of the whole layer, with the collectives behind `cut` ----------------------------------------------
class Saved:
    __slots__ = ('xt', 'mean', 'invstd', 'w', 'b', 'drop', 'relu', 'group', 'inv_count', 'inv_count_dev', 'pdtype')


def _group_of(norm):
    """The process group a synchronised layer talks to, or None for a plain (per-rank) BatchNorm / a one-rank job."""
    if not isinstance(norm, torch.nn.SyncBatchNorm) or not (dist.is_available() and dist.is_initialized()):
        return None
    return norm.process_group if norm.process_group is not None else dist.group.WORLD


def forward_pieces(xt, norm, relu, drop, cut=_call):
    """-> (y, Saved).  Updates the layer's running statistics / batch counter like the torch layers do."""
    C = xt.shape[-1]
    rows = xt.numel() // C
    group = _group_of(norm)
    track = norm.track_running_stats and norm.running_mean is not None
    if track and norm.num_batches_tracked is not None:
        norm.num_batches_tracked.add_(1)
    sv = Saved()
    if group is None:
        mean, invstd = local_stats(xt, norm.eps, norm.running_mean if track else None, norm.running_var if track else None, norm.momentum)
        sv.inv_count, sv.inv_count_dev = 1.0 / rows, None
    else:
        world = dist.get_world_size(group)
        mean_l, invstd_l = local_stats(xt, norm.eps)
        count = torch.full((1,), float(rows), dtype=torch.float32, device=xt.device)
        local = torch.cat([mean_l, invstd_l, count])
        gathered = torch.empty(world, 2 * C + 1, dtype=torch.float32, device=xt.device)
        if dist.get_backend(group) == 'gloo':     # no _allgather_base in gloo (torch's SyncBatchNorm makes the same distinction)
            cut(lambda: dist.all_gather(list(gathered.unbind(0)), local, group=group))
        else:
            cut(lambda: dist.all_gather_into_tensor(gathered, local, group=group))
        mean_all, invstd_all, count_all = torch.split(gathered, C, dim=1)
        counts = count_all.reshape(-1)
        mean, invstd = torch.batch_norm_gather_stats_with_counts(xt.new_empty(1, C, 1), mean_all, invstd_all,
                                                                 norm.running_mean if track else None, norm.running_var if track else None,
                                                                 norm.momentum, norm.eps, counts)
        sv.inv_count, sv.inv_count_dev = 0.0, counts.sum().reciprocal().reshape(1)   # ranks may hold unequal batches: total on device
    w, b = norm.weight.detach(), norm.bias.detach()
    y = apply_fwd(xt, mean, invstd, w, b, drop, relu)
    sv.xt, sv.mean, sv.invstd, sv.w, sv.b, sv.drop, sv.relu, sv.group, sv.pdtype = xt, mean, invstd, w, b, drop, relu, group, norm.weight.dtype
    return y, sv


def backward_pieces(sv, dy, need_dx=True, cut=_call):
    """-> (dx | None, grad_weight, grad_bias) -- the parameter gradients are THIS rank's (the data-parallel reducer averages them)."""
    dy = dy.contiguous()
    sums = bwd_reduce(sv.xt, dy, sv.mean, sv.invstd, sv.w, sv.b, sv.drop, sv.relu)
    gw, gb = (sums[1] * sv.invstd).to(sv.pdtype), sums[0].to(sv.pdtype, copy=True)   # copies: `sums` is summed over the ranks next
    if not need_dx:
        return None, gw, gb
    if sv.group is not None:
        cut(lambda: dist.all_reduce(sums, group=sv.group))
    dx = bwd_elemt(sv.xt, dy, sv.mean, sv.invstd, sv.w, sv.b, sv.drop, sv.relu, sums, sv.inv_count, sv.inv_count_dev)
    return dx, gw, gb


class _NormAct(torch.autograd.Function):
    """Eager form: the collectives (if any) are issued inline, like torch.nn.SyncBatchNorm's own autograd function."""

    @staticmethod
    def forward(ctx, xt, weight, bias, norm, relu, drop):
        y, sv = forward_pieces(xt, norm, relu, drop)
        ctx.sv = sv
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, gw, gb = backward_pieces(ctx.sv, dy, need_dx=ctx.needs_input_grad[0])
        return dx, gw if ctx.needs_input_grad[1] else None, gb if ctx.needs_input_grad[2] else None, None, None, None


def channel_dropout_scale(tokens, p):
    """Dropout2d on a [B, C, H, W] map = one Bernoulli(1-p)/(1-p) factor per (image, channel): [B, C] fp32."""
    keep = 1.0 - p
    return torch.empty(tokens.shape[0], tokens.shape[-1], dtype=torch.float32, device=tokens.device).bernoulli_(keep).div_(keep)


def norm_act(tokens, norm, relu=True, drop=None):
    """y = act(BatchNorm_train(tokens)) * drop: tokens [B, N, C]; `norm` a BatchNorm2d / SyncBatchNorm in training mode."""
    rec = getattr(norm, '_segments', None)
    if rec is not None:
        return rec.fused_norm_act(norm, tokens, relu, drop)
    return _NormAct.apply(tokens, norm.weight, norm.bias, norm, relu, drop)

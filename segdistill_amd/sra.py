"""autograd binding of the spatial-reduction attention kernels (csrc/sra_attn.hip)."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _DT, _stream_ptr


def supported(q, kv, heads):
    """q [B, N, C], kv [B, KV, 2C] -- the q / kv Linear outputs; C = heads * head_dim with head_dim in {32, 64}; at most 256
    keys (K and V of a head live in LDS)."""
    if not (q.is_cuda and q.dtype in _DT and kv.dtype == q.dtype and q.dim() == 3 and kv.dim() == 3):
        return False
    C = q.shape[-1]
    return (heads > 0 and C % heads == 0 and C // heads in (32, 64) and kv.shape[-1] == 2 * C and kv.shape[0] == q.shape[0]
            and q.shape[1] > 0 and 0 < kv.shape[1] <= 256)


def preferred(n_queries, n_keys, head_dim, q):
    """Policy, from tools/sra_bench.py on MI355X (profiles/r01_step_kernels_microbench.txt, profiles/r02_sra_bench_bf16.txt): the kernels
    beat both library forms (fused SDPA, bmm+softmax) at every MiT stage shape, forward and forward+backward -- fp32 storage since round 1
    (exact f32 MFMA, now split-bf16), bf16 storage since the bf16-MFMA kernels of round 2 (sra_fwd_b16 / sra_bwd_dq_b16 / sra_bwd_dkv_b16:
    e.g. 8 x 16384 queries, head_dim 64: forward 33 vs 39 us, forward + backward 176 vs 628 us, eager).
    SEGDISTILL_SRA=off|train|all overrides (benchmarking / bisecting)."""
    import os
    mode = os.environ.get('SEGDISTILL_SRA', 'auto')
    training = torch.is_grad_enabled() and q.requires_grad
    if mode == 'off':
        return False
    if mode == 'all':
        return True
    if mode == 'train':
        return training
    return True


class _SRAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        q, kv = q.contiguous(), kv.contiguous()
        B, N, C = q.shape
        KV, D = kv.shape[1], C // heads
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, N, dtype=torch.float32, device=q.device)
        rc = _lib.lib().sd_sra_fwd(q.data_ptr(), kv.data_ptr(), out.data_ptr(), lse.data_ptr(), _DT[q.dtype], B, N, KV, heads, D, float(scale),
                                   _stream_ptr())
        _lib.check(rc, 'sd_sra_fwd')
        ctx.save_for_backward(q, kv, out, lse)
        ctx.heads, ctx.scale = heads, float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse = ctx.saved_tensors
        dout = dout.contiguous()
        B, N, C = q.shape
        KV, heads = kv.shape[1], ctx.heads
        D = C // heads
        L = _lib.lib()
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        wsb = L.sd_sra_workspace_bytes(B, N, KV, heads, D)
        ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
        rc = L.sd_sra_bwd(q.data_ptr(), kv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dq.data_ptr(), dkv.data_ptr(),
                          _DT[q.dtype], B, N, KV, heads, D, ctx.scale, ws.data_ptr(), wsb, _stream_ptr())
        _lib.check(rc, 'sd_sra_bwd')
        return dq, dkv, None, None


def sr_attention(q, kv, heads, scale):
    """softmax(scale * q k^T) v per head; q [B,N,C], kv [B,KV,2C] (k | v, head-major inside) -> [B,N,C]."""
    return _SRAttention.apply(q, kv, heads, scale)

#!/usr/bin/env python3
"""Spatial-reduction attention: csrc/sra_attn.hip vs the library paths (fused SDPA, explicit bmm+softmax) at the MiT stage
shapes of a 512x512 input, batch 8, fp32.  HIP-event timing, median of 20."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from segdistill_amd import sra

dev = torch.device('cuda:0')


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def lib_attention(q, kv, heads, scale, explicit):
    B, N, C = q.shape
    d = C // heads
    qh = q.reshape(B, N, heads, d).transpose(1, 2)
    kvh = kv.reshape(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4)
    if explicit:
        out = ((qh @ kvh[0].transpose(-2, -1)) * scale).softmax(dim=-1) @ kvh[1]
    else:
        out = F.scaled_dot_product_attention(qh, kvh[0], kvh[1], scale=scale)
    return out.transpose(1, 2).reshape(B, N, C)


def main():
    dtype = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == 'bf16' else torch.float32
    print(f'# dtype {dtype}; us per call (median of 20)')
    print(f'{"shape (B,N,KV,heads,D)":28s} {"hip fwd":>9s} {"sdpa fwd":>9s} {"expl fwd":>9s} | {"hip f+b":>9s} {"sdpa f+b":>9s} {"expl f+b":>9s}')
    for (heads, D, tag) in ((None, 32, 'B0'), (None, 64, 'B1-B5')):
        for N, hd in ((16384, 1), (4096, 2), (1024, 5), (256, 8)):
            B, KV = 8, 256
            C = hd * D
            q = torch.randn(B, N, C, device=dev, dtype=dtype, requires_grad=True)
            kv = torch.randn(B, KV, 2 * C, device=dev, dtype=dtype, requires_grad=True)
            do = torch.randn(B, N, C, device=dev, dtype=dtype)
            scale = D ** -0.5
            res = []
            with torch.no_grad():
                res.append(timeit(lambda: sra.sr_attention(q, kv, hd, scale)))
                res.append(timeit(lambda: lib_attention(q, kv, hd, scale, False)))
                res.append(timeit(lambda: lib_attention(q, kv, hd, scale, True)))

            def fb(fn):
                def run():
                    q.grad = kv.grad = None
                    fn().backward(do)
                return run
            res.append(timeit(fb(lambda: sra.sr_attention(q, kv, hd, scale))))
            res.append(timeit(fb(lambda: lib_attention(q, kv, hd, scale, False))))
            res.append(timeit(fb(lambda: lib_attention(q, kv, hd, scale, True))))
            print(f'{tag + " " + str((B, N, KV, hd, D)):28s} ' + ' '.join(f'{r:9.1f}' for r in res[:3]) + ' | ' + ' '.join(f'{r:9.1f}' for r in res[3:]))


if __name__ == '__main__':
    main()

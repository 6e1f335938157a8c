"""The MiT patch-embedding projections of config 2 (fp32; student B0, teacher B2) as window gather + token GEMM (csrc/patch_embed.hip) next to
nn.Conv2d through MIOpen: device time inside a replayed graph, forward alone and forward + backward.   python tools/patch_embed_bench.py [--bf16]"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import timeit  # noqa: E402

from segdistill_amd import deferred, patch_embed  # noqa: E402

dev = torch.device('cuda:0')
bf16 = '--bf16' in sys.argv
torch.backends.cudnn.benchmark = True
# (tag, B, cin, cout, H, W, k, s, p)
SHAPES = [('B0 pe1', 8, 3, 32, 512, 512, 7, 4, 3), ('B0 pe2', 8, 32, 64, 128, 128, 3, 2, 1), ('B0 pe3', 8, 64, 160, 64, 64, 3, 2, 1),
          ('B0 pe4', 8, 160, 256, 32, 32, 3, 2, 1), ('B2 pe1', 8, 3, 64, 512, 512, 7, 4, 3), ('B2 pe2', 8, 64, 128, 128, 128, 3, 2, 1),
          ('B2 pe3', 8, 128, 320, 64, 64, 3, 2, 1), ('B2 pe4', 8, 320, 512, 32, 32, 3, 2, 1)]
tot = [0.0, 0.0, 0.0, 0.0]
for tag, B, cin, cout, H, W, k, s, p in SHAPES:
    conv = nn.Conv2d(cin, cout, k, s, p).to(dev)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    if cin == 3:
        x = torch.randn(B, cin, H, W, device=dev)
    else:
        x = torch.randn(B, H * W, cin, device=dev).reshape(B, H, W, cin).permute(0, 3, 1, 2)
    xg = x.detach().clone().requires_grad_(cin != 3) if cin != 3 else x
    if cin != 3:
        xg = torch.randn(B, H * W, cin, device=dev).requires_grad_(True)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    g_tok = torch.randn(B, Ho * Wo, cout, device=dev)

    def view(t):
        return t if cin == 3 else t.reshape(B, H, W, cin).permute(0, 3, 1, 2)

    def ours_fwd():
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            return patch_embed.patch_embed_tokens(view(xg.detach() if cin != 3 else x), conv)[0]

    def lib_fwd():
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            return conv(view(xg.detach() if cin != 3 else x))

    def ours_fb():
        for q in conv.parameters():
            q.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            y = patch_embed.patch_embed_tokens(view(xg), conv)[0]
        with deferred.scope():
            y.backward(g_tok.to(y.dtype))

    def lib_fb():
        for q in conv.parameters():
            q.grad = None
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
            y = conv(view(xg))
        y.permute(0, 2, 3, 1).reshape(B, Ho * Wo, cout).backward(g_tok.to(y.dtype))

    t = [timeit(f, 10, per_graph=5) for f in (ours_fwd, lib_fwd, ours_fb, lib_fb)]
    for i in range(4):
        tot[i] += t[i]
    print(f'{tag}  {cin:>3} -> {cout:>3}  {H}x{W} k{k} s{s}:  fwd ours {t[0]:7.1f} us  MIOpen {t[1]:7.1f} us   |  fwd+bwd ours {t[2]:7.1f} us  MIOpen {t[3]:7.1f} us')
print(f'sum: fwd ours {tot[0]:.1f} MIOpen {tot[1]:.1f} | fwd+bwd ours {tot[2]:.1f} MIOpen {tot[3]:.1f}')

"""Depthwise 3x3 (+ bias + GELU) of the Mix-FFN at the Segformer stage shapes of BASELINE configs 2 / 5 (B0/B1 student, B2/B4 teacher; token
layout [B, H*W, 4*dim]): device time per launch from a replayed hipGraph against the HBM floor (one read + one write of the map at 6.3 TB/s;
training also writes the pre-activation).

    python tools/dw_bench.py [--dtype f32|bf16] [--B 8] [--shape side,C]      (one shape only: for rocprofv3 --pmc passes)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from segdistill_amd.dwconv import dwconv3x3_gelu_tokens, dwconv3x3_gelu_tokens_inference  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--B', type=int, default=8)
    ap.add_argument('--shape', default=None)
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    dev = torch.device('cuda:0')
    print(f'{"map":>12} {"C":>5} | {"frozen us":>9} {"floor":>6} | {"train fwd":>9} {"floor":>6} | {"fwd+bwd":>8}')
    shapes = [(side, C) for side, dims in ((128, (128, 256)), (64, (256, 512)), (32, (640, 1280)), (16, (1024, 2048))) for C in dims]
    if a.shape:
        shapes = [tuple(int(v) for v in a.shape.split(','))]
    for side, C in shapes:
        if True:
            x = torch.randn(a.B, side * side, C, device=dev, dtype=dt, requires_grad=True)
            w = torch.randn(C, 1, 3, 3, device=dev, requires_grad=True)
            b = torch.randn(C, device=dev, requires_grad=True)
            nbytes = x.numel() * x.element_size()
            with torch.no_grad():
                f = timeit(lambda: dwconv3x3_gelu_tokens_inference(x, w, b, side, side))
            with torch.no_grad():
                dy = torch.randn_like(x)
            t = timeit(lambda: dwconv3x3_gelu_tokens(x, w, b, side, side))
            fb = timeit(lambda: torch.autograd.grad(dwconv3x3_gelu_tokens(x, w, b, side, side), (x, w, b), dy))
            print(f'{a.B}x{side}x{side:<5} {C:>5} | {f:9.1f} {2 * nbytes / 6.3e6:6.1f} | {t:9.1f} {3 * nbytes / 6.3e6:6.1f} | {fb:8.1f}')


if __name__ == '__main__':
    main()

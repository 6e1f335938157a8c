"""LayerNorm forward / backward at the Segformer stage shapes of BASELINE configs 2 and 5 (run on the GPU box): device time per launch
from a replayed hipGraph, against the HBM floor of each pass (fwd 2 tensors, bwd 3 tensors of rows x C elements at 6.3 TB/s).

    python tools/ln_bench.py [--dtype f32|bf16] [--B 8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from segdistill_amd.layernorm import HipLayerNorm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--B', type=int, default=8)
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    dev = torch.device('cuda:0')
    print(f'{"rows":>8} {"C":>5} | {"fwd us":>8} {"floor":>6} | {"bwd us":>8} {"floor":>6}')
    for side, dims in ((128, (32, 64)), (64, (64, 128)), (32, (160, 320)), (16, (256, 512))):
        for C in dims:
            rows = a.B * side * side
            ln = HipLayerNorm(C).to(dev)
            x = torch.randn(rows, C, device=dev, dtype=dt, requires_grad=True)
            dy = torch.randn(rows, C, device=dev, dtype=dt)
            e = x.element_size()
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
                f = timeit(lambda: ln(x))
                # the backward runs on the stream its forward ran on, so both are captured together and the forward is subtracted
                b = timeit(lambda: torch.autograd.grad(ln(x), x, dy)) - f
            print(f'{rows:>8} {C:>5} | {f:8.1f} {2 * rows * C * e / 6.3e6:6.1f} | {b:8.1f} {3 * rows * C * e / 6.3e6:6.1f}')


if __name__ == '__main__':
    main()

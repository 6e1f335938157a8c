"""One token-major Linear product, launched N times, for rocprofv3 passes (kernel trace or --pmc) over a single GEMM kernel.

    python tools/gemm_one.py --T 131072 --K 256 --N 256 --mode pl|x3|f32|lib [--dir fwd|bwd] [--reps 6]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=131072)
    ap.add_argument('--K', type=int, default=256)
    ap.add_argument('--N', type=int, default=256)
    ap.add_argument('--mode', default='pl')
    ap.add_argument('--dir', default='fwd')
    ap.add_argument('--reps', type=int, default=6)
    a = ap.parse_args()
    from segdistill_amd import _lib, planes, token_gemm
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    x = torch.randn(a.T, a.K if a.dir == 'fwd' else a.N, device=dev)
    w = torch.randn(a.N, a.K, device=dev) * 0.05
    b = torch.randn(a.N, device=dev)
    if a.mode == 'pl':
        p = planes.get(w, a.dir)
        fn = (lambda: token_gemm.linear_fwd_planes(x, w, p, b)) if a.dir == 'fwd' else (lambda: token_gemm.linear_bwd_data_planes(x, w, p))
    elif a.mode == 'lib':
        fn = (lambda: torch.nn.functional.linear(x, w, b)) if a.dir == 'fwd' else (lambda: x @ w)
    else:
        sp = a.mode == 'x3'
        fn = (lambda: token_gemm.linear_fwd(x, w, b, split_bf16=sp)) if a.dir == 'fwd' else (lambda: token_gemm.linear_bwd_data(x, w, split_bf16=sp))
    for _ in range(a.reps):
        y = fn()
    torch.cuda.synchronize()
    print('ok', tuple(y.shape), float(y.float().abs().mean()))


if __name__ == '__main__':
    main()

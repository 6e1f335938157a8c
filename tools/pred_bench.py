"""linear_pred of the frozen teacher head at BASELINE config 2 (tokens [8, 16384, 768] -> 150 class planes, fp32): the library's batched
W x tokens^T against the class-plane kernel (sd_linear_nchw_fwd).  Device time per call from a replayed hipGraph.

    python tools/pred_bench.py [--E 768] [--classes 150]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from segdistill_amd import _lib  # noqa: E402
from segdistill_amd import linear as _linear  # noqa: E402
from segdistill_amd.linear import linear_to_planes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--E', type=int, default=768)
    ap.add_argument('--classes', type=int, default=150)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    for B, P in ((8, 16384), (8, 4096)):
      for E in sorted({a.E, 256}):
        x = torch.randn(B, P, E, device=dev)
        w = torch.randn(a.classes, E, device=dev) * 0.05
        b = torch.randn(a.classes, device=dev)
        with torch.no_grad():
            lib = timeit(lambda: torch.baddbmm(b.view(1, -1, 1), w.unsqueeze(0).expand(B, -1, -1), x.transpose(1, 2)))
            hip_pl = timeit(lambda: linear_to_planes(x, w, b))     # 160-row tile on pre-split row planes (round-3 A/B arms: profiles/r03_pred_bench.txt)
            ref = torch.baddbmm(b.view(1, -1, 1).double(), w.double().unsqueeze(0).expand(B, -1, -1), x.double().transpose(1, 2))
            e_lib = (torch.baddbmm(b.view(1, -1, 1), w.unsqueeze(0).expand(B, -1, -1), x.transpose(1, 2)).double() - ref).abs().max().item()
            e_hip = (linear_to_planes(x, w, b).double() - ref).abs().max().item()
        gf = 2.0 * B * P * E * a.classes / 1e9
        print(f'B={B} P={P} E={E} -> {a.classes}: library {lib:7.1f} us ({gf / lib * 1e3:5.1f} TF, max err {e_lib:.2e}) | class-plane kernel, 160-row tile on '
              f'pre-split W planes {hip_pl:7.1f} us ({gf / hip_pl * 1e3:5.1f} TF, max err {e_hip:.2e})')


if __name__ == '__main__':
    main()

"""SegFormer-head up-sample + sum (csrc/headfuse.hip, upsum_fwd_strip) at BASELINE config 2 / 5 shapes: device time per launch from a replayed
hipGraph against the HBM floor (the full-resolution branch in and out plus the three coarse branches, at 6.3 TB/s).

    python tools/upsum_bench.py [--B 8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from segdistill_amd.headfuse import upsum  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=8)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    sizes = [(128, 128), (64, 64), (32, 32), (16, 16)]
    for dt in (torch.float32, torch.bfloat16):
        for E in (768, 256):
            zs = [torch.randn(a.B, h * w, E, device=dev).to(dt) for h, w in sizes]
            with torch.no_grad():
                us = timeit(lambda: upsum(zs[0], zs[1], zs[2], zs[3], None, sizes))
            nbytes = (2 * zs[0].numel() + sum(z.numel() for z in zs[1:])) * zs[0].element_size()
            print(f'upsum_fwd {str(dt)[6:]:>8} E={E:<4} {us:7.1f} us   floor {nbytes / 6.3e6:6.1f} us   {nbytes / us / 1e6:5.2f} TB/s')


if __name__ == '__main__':
    main()

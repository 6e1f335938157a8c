"""The fused tail of the frozen SegFormer head (csrc/head_tail.hip) against the two kernels it replaces (sum + norm + ReLU, then linear_pred) at
the teacher's shape (8 x 128 x 128, E = 768, 150 classes); device time per call from a replayed hipGraph.   python tools/head_tail_bench.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from segdistill_amd import headfuse  # noqa: E402
from segdistill_amd.linear import linear_to_planes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=8)
    ap.add_argument('--only-fused', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    print(f'{"map":>12} {"E":>4} | {"sum+norm":>8} {"pred":>7} {"sum":>7} | {"fused":>7} | MFMA floor (6 bf16 products at 2.5 PF/s), HBM floor (z1 in + logits out at 6.3 TB/s)')
    for side, E in ((128, 768), (128, 256)):
        sizes = [(side, side), (side // 2, side // 2), (side // 4, side // 4), (side // 8, side // 8)]
        with torch.no_grad():
            zs = [torch.randn(a.B, h * w, E, device=dev) for (h, w) in sizes]
            scale, shift = torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev)
            wp = (torch.randn(150, E, device=dev) / E ** 0.5).requires_grad_(False)
            bp = torch.randn(150, device=dev)
            t_f = timeit(lambda: headfuse.head_tail(zs, sizes, None, scale, shift, wp, bp))
            t_s = t_p = float('nan')
            if not a.only_fused:
                y = headfuse.upsum_affine_inference(zs, None, sizes, scale, shift, relu=True)
                t_s = timeit(lambda: headfuse.upsum_affine_inference(zs, None, sizes, scale, shift, relu=True))
                t_p = timeit(lambda: linear_to_planes(y, wp, bp))
        mf = a.B * side * side * 160 * E * 2 * 6 / 2.5e9
        hb = (zs[0].numel() + a.B * 150 * side * side) * 4 / 6.3e6
        print(f'{a.B}x{side}x{side:<5} {E:>4} | {t_s:8.1f} {t_p:7.1f} {t_s + t_p:7.1f} | {t_f:7.1f} | {mf:6.1f} {hb:6.1f}')


def stamps():
    # a -DSD_HEAD_TAIL_STAMPS build: make -C segdistill_amd/csrc OUTDIR=../lib_ab EXTRA=-DSD_HEAD_TAIL_STAMPS; SEGDISTILL_LIB=.../lib_ab/libsegdistill_hip.so
    import ctypes as C
    from segdistill_amd import _lib
    raw = C.CDLL(_lib.LIB_PATH)
    dev = torch.device('cuda:0')
    side, E, B = 128, 768, 8
    sizes = [(side, side), (side // 2, side // 2), (side // 4, side // 4), (side // 8, side // 8)]
    with torch.no_grad():
        zs = [torch.randn(B, h * w, E, device=dev) for (h, w) in sizes]
        scale, shift = torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev)
        wp, bp = torch.randn(150, E, device=dev) / E ** 0.5, torch.randn(150, device=dev)
        for _ in range(3):
            headfuse.head_tail(zs, sizes, None, scale, shift, wp, bp)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert raw.sd_debug_head_tail_stamps(buf) == 0
    n = max(1, buf[9])
    names = ['W chunk DMA issued', 'sum stage', 'requests issued', 'affine + split + LDS stores', 'wait for the W chunk', 'barrier 1', 'fragment reads + MFMAs',
             'wait for operands + barrier 2']
    print(f'wave 0 of workgroup 0, {n} chunks, s_memtime ticks (100 MHz): total {buf[8]}')
    for i, nm in enumerate(names):
        print(f'  {nm:32s} {buf[i]:8d} ticks  ({buf[i] / n:8.1f} per chunk)')


if __name__ == '__main__':
    stamps() if '--stamps' in sys.argv else main()

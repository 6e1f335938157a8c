"""The fused Mix-FFN tail (csrc/mixffn_tail.hip) against the two kernels it replaces (dwconv + GELU, then fc2) at the frozen teacher's stage shapes;
device time per call from a replayed hipGraph.    python tools/mixffn_tail_bench.py [--B 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import timeit  # noqa: E402
from segdistill_amd import dwconv as hip_dw, mixffn  # noqa: E402
from segdistill_amd.backbones.mit import MixFFN  # noqa: E402
from segdistill_amd.linear import call_linear  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=8)
    ap.add_argument('--dtype', default='f32')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    print(f'{"map":>12} {"dim":>4} | {"dw+gelu":>8} {"fc2":>7} {"sum":>7} | {"fused":>7} | HBM floor (h in + y out at 6.3 TB/s)')
    for side, dim in ((128, 64), (64, 128), (32, 64), (32, 128)):
        m = MixFFN(dim, 4 * dim).to(dev).eval()
        conv = m.dwconv.dwconv
        amp = torch.autocast('cuda', dtype=torch.bfloat16, enabled=a.dtype == 'bf16')
        with torch.no_grad(), amp:
            h = torch.randn(a.B, side * side, 4 * dim, device=dev)
            h = h.bfloat16() if a.dtype == 'bf16' else h
            g = hip_dw.dwconv3x3_gelu_tokens_inference(h, conv.weight, conv.bias, side, side)
            t_dw = timeit(lambda: hip_dw.dwconv3x3_gelu_tokens_inference(h, conv.weight, conv.bias, side, side))
            t_fc = timeit(lambda: call_linear(m.fc2, g))
            t_f = timeit(lambda: mixffn.tail(h, conv, m.fc2, (side, side)))
        floor = (h.numel() + h.numel() // 4) * h.element_size() / 6.3e6
        print(f'{a.B}x{side}x{side:<5} {dim:>4} | {t_dw:8.1f} {t_fc:7.1f} {t_dw + t_fc:7.1f} | {t_f:7.1f} | {floor:6.1f}')


def stamps():
    # a -DSD_MIXFFN_TAIL_STAMPS build: make -C segdistill_amd/csrc OUTDIR=../lib_ab EXTRA=-DSD_MIXFFN_TAIL_STAMPS; SEGDISTILL_LIB=.../lib_ab/libsegdistill_hip.so
    import ctypes as C
    from segdistill_amd import _lib
    raw = C.CDLL(_lib.LIB_PATH)
    dev = torch.device('cuda:0')
    for side, dim in ((128, 64), (64, 128)):
        m = MixFFN(dim, 4 * dim).to(dev).eval()
        with torch.no_grad():
            h = torch.randn(8, side * side, 4 * dim, device=dev)
            for _ in range(3):
                mixffn.tail(h, m.dwconv.dwconv, m.fc2, (side, side))
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 16)()
        assert raw.sd_debug_mixffn_tail_stamps(buf) == 0
        n = max(1, buf[9])
        names = ['taps + convolution', 'weight chunk split + stores', 'requests issued', 'GELU + split + LDS stores', 'barrier 1', 'fragment reads + MFMAs',
                 'wait for operands', 'barrier 2']
        print(f'8x{side}x{side} dim {dim}: wave 0 of workgroup 0, {n} chunks, s_memtime ticks: total {buf[8]}')
        for i, nm in enumerate(names):
            print(f'  {nm:32s} {buf[i]:8d} ticks  ({buf[i] / n:8.1f} per chunk)')


if __name__ == '__main__':
    stamps() if '--stamps' in sys.argv else main()

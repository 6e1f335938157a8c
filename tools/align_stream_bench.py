"""The NCHW align projection's forward (config 4: [8,128,64,64] -> 512 channels): csrc/align_stream.hip (W resident in registers) vs the generic
pipelined GEMM it replaces (tunable align_stream = 0), device time inside a replayed graph.   python tools/align_stream_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import timeit  # noqa: E402

from segdistill_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda:0')
for (B, Cs, Ct, h, w) in [(8, 128, 512, 64, 64), (8, 64, 256, 128, 128), (8, 128, 512, 128, 128), (2, 128, 512, 64, 64)]:
    x = torch.randn(B, Cs, h, w, device=dev)
    wt = torch.randn(Ct, Cs, device=dev) / Cs ** 0.5
    b = torch.randn(Ct, device=dev)
    y = torch.empty(B, Ct, h, w, device=dev)

    def fwd():
        rc = L.sd_align1x1_fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), 0, B, Cs, Ct, h, w, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    res = {}
    for flag in (1, 0, 1):
        L.sd_set_tunable(b'align_stream', flag)
        res.setdefault(flag, []).append(timeit(fwd, 20))
    L.sd_set_tunable(b'align_stream', 1)
    P = h * w
    nbytes = (B * (Cs + Ct) * P + Ct * Cs) * 4
    flops = 2.0 * B * Ct * Cs * P
    t1 = min(res[1])
    print(f'[{B},{Cs},{h},{w}] -> {Ct}: stream {res[1][0]:6.1f} / {res[1][1]:6.1f} us   generic {res[0][0]:6.1f} us   '
          f'stream: {nbytes / t1 / 1e3:7.1f} GB/s = {100 * nbytes / t1 / 1e3 / 8000:4.1f} % of HBM, {flops / t1 / 1e6:6.1f} TFLOP/s = {100 * flops / t1 / 1e6 / 417:4.1f} % of bf16/6')

if '--stamps' in sys.argv:       # a -DSD_ALIGN_STREAM_STAMPS build (make OUTDIR=../lib_ab EXTRA=-DSD_ALIGN_STREAM_STAMPS; SEGDISTILL_LIB=...)
    import ctypes as C
    raw = C.CDLL(_lib.LIB_PATH)
    B, Cs, Ct, h, w = 8, 128, 512, 128, 128
    x = torch.randn(B, Cs, h, w, device=dev)
    wt = torch.randn(Ct, Cs, device=dev) / Cs ** 0.5
    y = torch.empty(B, Ct, h, w, device=dev)
    for _ in range(3):
        L.sd_align1x1_fwd(x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, B, Cs, Ct, h, w, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    assert raw.sd_debug_align_stream_stamps(buf) == 0
    n = max(1, buf[8])
    names = ['prologue (W + first request)', 'wait for operands', 'split + LDS stores', 'barrier 1', 'requests + reads + MFMAs', 'stores issued', 'barrier 2']
    print(f'wave 0 of workgroup 0, {n} tiles, s_memtime ticks: total {buf[7]}')
    for i, nm in enumerate(names):
        print(f'  {nm:32s} {buf[i]:8d} ticks  ({buf[i] / (n if i else 1):8.1f} per tile)' if i else f'  {nm:32s} {buf[i]:8d} ticks')

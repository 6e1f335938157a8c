"""Diagnostic for VERDICT r3 item 2 ("split each activation ONCE, at its producer"): how much of the planes GEMM's time IS the in-loop split
of the activation operand?  Times sd_linear_fwd_planes on the big token products of config 2 with the product library and with a
-DSD_DIAG_NOSPLIT build of the same library (one conversion per value pair, no residual arithmetic -- wrong numerics, same memory traffic,
same MFMA count): the difference is an UPPER BOUND on what a pre-split activation can buy inside the GEMM, before the producers' extra
write traffic (6 instead of 4 bytes per element) is charged.
    make -C segdistill_amd/csrc OUTDIR=../lib_ab EXTRA=-DSD_DIAG_NOSPLIT
    python tools/gemm_nosplit_probe.py ; SEGDISTILL_LIB=$PWD/segdistill_amd/lib_ab/libsegdistill_hip.so python tools/gemm_nosplit_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import timeit  # noqa: E402

from segdistill_amd import _lib, planes, token_gemm  # noqa: E402

SHAPES = [(131072, 256, 256), (131072, 64, 256), (131072, 256, 64), (131072, 64, 64), (32768, 128, 512), (32768, 512, 128), (32768, 128, 128),
          (8192, 320, 1280), (8192, 1280, 320), (8192, 320, 320), (131072, 256, 768)]
dev = torch.device('cuda:0')
print('library:', _lib.LIB_PATH)
tot = 0.0
for T, K, N in SHAPES:
    x = torch.randn(T, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    pf = planes.get(w, 'fwd')
    t = timeit(lambda: token_gemm.linear_fwd_planes(x, w, pf, b), 20)
    tot += t
    print(f'{T:>7} x {K:>5} -> {N:>5}: {t:8.1f} us')
print(f'sum {tot:8.1f} us')

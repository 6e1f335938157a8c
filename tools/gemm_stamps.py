"""Where does a k-step of the planes GEMM go?  Runs sd_linear_fwd_planes from a DIAGNOSTIC build of the library
(make -C segdistill_amd/csrc OUTDIR=../lib_ab EXTRA=-DSD_GEMM_STAMPS; SEGDISTILL_LIB=.../lib_ab/libsegdistill_hip.so) that stamps s_memtime at
the phase boundaries of every k-step in wave 0 of every 64th workgroup, and prints the per-phase cycles per k-step (median over workgroups).

phases: 0 wait B(kt,0) | 1 split A half 1 + MFMAs half 0 | 2 wait A(kt+1) | 3 LDS store + requests + barrier | 4 wait B(kt,1) | 5 split + MFMAs half 1
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=131072)
    ap.add_argument('--K', type=int, default=256)
    ap.add_argument('--N', type=int, default=256)
    a = ap.parse_args()
    from segdistill_amd import _lib, planes, token_gemm
    L = _lib.lib()
    fn = getattr(L, 'sd_debug_gemm_stamps', None)
    assert fn is not None, 'needs the -DSD_GEMM_STAMPS build (SEGDISTILL_LIB)'
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    x = torch.randn(a.T, a.K, device=dev)
    w = torch.randn(a.N, a.K, device=dev) * 0.05
    b = torch.randn(a.N, device=dev)
    p = planes.get(w, 'fwd')
    tiles = -(-a.T // 128) * -(-a.N // 128)
    buf = torch.zeros((tiles // 64 + 1) * 12, dtype=torch.int64, device=dev)
    fn(buf.data_ptr())
    for _ in range(5):
        token_gemm.linear_fwd_planes(x, w, p, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    token_gemm.linear_fwd_planes(x, w, p, b)
    e1.record()
    torch.cuda.synchronize()
    r = buf.view(-1, 12).cpu()
    r = r[r[:, 7] > 0]
    names = ['wait B(kt,0)', 'split + MFMA half 0', 'wait A(kt+1)', 'store + requests + barrier', 'wait B(kt,1)', 'split + MFMA half 1']
    print(f'T={a.T} K={a.K} N={a.N}: {len(r)} stamped workgroups, {int(r[0, 7])} pipelined k-steps each; cycles per k-step (median / min / max over workgroups)')
    tot = 0.0
    for i, nme in enumerate(names):
        v = (r[:, i].double() / r[:, 7].double())
        tot += float(v.median())
        print(f'  phase {i} {nme:<28} {float(v.median()):8.0f} {float(v.min()):8.0f} {float(v.max()):8.0f}')
    whole = r[:, 6].double()
    print(f'  sum of phase medians {tot:8.0f} per k-step; whole main loop {float(whole.median()):.0f} cycles (median); MFMA floor 48 x 32 = 1536 per k-step per wave')


    # the last launch as a timeline: when sampled workgroups entered / left, in s_memtime ticks from the first entry
    ent, end = r[:, 9].double(), r[:, 10].double()
    t0 = float(ent.min())
    span = float(end.max()) - t0
    us = e0.elapsed_time(e1) * 1e3
    print(f'  kernel {us:.1f} us between events; sampled span {span:.0f} ticks -> {span / us:.0f} ticks/us if the span were the whole kernel')
    pro, epi = r[:, 8].double(), (end - ent - r[:, 8].double() - whole)
    print(f'  per workgroup (median): prologue {float(pro.median()):.0f}, main loop {float(whole.median()):.0f}, last k-step + epilogue {float(epi.median()):.0f}, '
          f'lifetime {float((end - ent).median()):.0f} ticks')
    order = torch.argsort(ent)
    starts = ((ent[order] - t0) / max(span, 1) * 100).tolist()
    ends = ((end[order] - t0) / max(span, 1) * 100).tolist()
    print('  entry -> exit of sampled workgroups, % of the span: ' + ' '.join(f'{a_:.0f}-{b_:.0f}' for a_, b_ in zip(starts[:32], ends[:32])))


if __name__ == '__main__':
    main()

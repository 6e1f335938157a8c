"""Kernel-level bench of the CGD criterion at BASELINE config-2 shapes (run on the GPU box).

usage: python tools/kbench.py [--B 8] [--C 150] [--HW 512] [--g 8] [--iters 1,2,4,8,16,32] [--dtype f32|bf16]
Reports HIP-event time per launch and algorithmic GB/s (fwd 2*N*e, bwd 3*N*e)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import _lib, ops  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=8)
    ap.add_argument('--C', type=int, default=150)
    ap.add_argument('--HW', type=int, default=512)
    ap.add_argument('--g', type=int, default=8)
    ap.add_argument('--tau', type=float, default=4.0)
    ap.add_argument('--iters', default='1,2,4,8,16,32')
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--nt', type=int, default=1)
    ap.add_argument('--bwd-unroll', type=int, default=4)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    torch.manual_seed(1234)
    S = (2 * torch.randn(a.B, a.C, a.HW, a.HW, device=dev)).to(dt)
    T = (2 * torch.randn(a.B, a.C, a.HW, a.HW, device=dev)).to(dt)
    N = S.numel()
    e = S.element_size()
    print(f'shape {tuple(S.shape)} {a.dtype} N={N} fwd bytes {2*N*e/1e9:.3f} GB bwd bytes {3*N*e/1e9:.3f} GB')
    # raw-ABI launches (no autograd overhead)
    L = _lib.lib()
    G = -(-a.C // a.g)
    rows = a.B * G
    row_lse2 = torch.empty(rows, 2, device=dev)
    row_kl = torch.empty(rows, device=dev)
    loss = torch.empty((), device=dev)
    dS = torch.empty_like(S)
    up = torch.ones((), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    DT = {torch.float32: 0, torch.bfloat16: 1}[dt]
    for it in [int(x) for x in a.iters.split(',')]:
        _lib.set_tunable('cgd_fwd_chunk_iters', it)
        _lib.set_tunable('cgd_bwd_chunk_iters', it)
        _lib.set_tunable('cgd_bwd_nt_store', a.nt)
        _lib.set_tunable('cgd_bwd_unroll', a.bwd_unroll)
        wsb = L.sd_cgd_kl_workspace_bytes(a.B, a.C, a.HW, a.HW, a.g)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

        def fwd():
            rc = L.sd_cgd_kl_fwd(S.data_ptr(), T.data_ptr(), DT, a.B, a.C, a.HW, a.HW, a.g, 1 / a.tau, 3.0 / rows, None,
                                 row_lse2.data_ptr(), row_kl.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, st)
            assert rc == 0, rc

        def bwd():
            rc = L.sd_cgd_kl_bwd(S.data_ptr(), T.data_ptr(), DT, a.B, a.C, a.HW, a.HW, a.g, 1 / a.tau, 3.0 / (rows * a.tau), None,
                                 row_lse2.data_ptr(), up.data_ptr(), dS.data_ptr(), st)
            assert rc == 0, rc

        tf, tfmin = timeit(fwd)
        tb, tbmin = timeit(bwd)
        print(f'chunk_iters {it:3d}: fwd {tf:.3f} ms ({2*N*e/tf/1e6:.0f} GB/s, best {2*N*e/tfmin/1e6:.0f})  '
              f'bwd {tb:.3f} ms ({3*N*e/tb/1e6:.0f} GB/s, best {3*N*e/tbmin/1e6:.0f})  '
              f'fwd+bwd {tf+tb:.3f} ms ({5*N*e/(tf+tb)/1e6:.0f} GB/s)  loss {float(loss):.6f}')
    # reference points on the same box: torch copy / add
    Y = torch.empty_like(S)
    tc, _ = timeit(lambda: Y.copy_(S))
    print(f'torch copy_: {tc:.3f} ms ({2*N*e/tc/1e6:.0f} GB/s)')
    ta, _ = timeit(lambda: torch.add(S, T, out=Y))
    print(f'torch add  : {ta:.3f} ms ({3*N*e/ta/1e6:.0f} GB/s)')


if __name__ == '__main__':
    main()

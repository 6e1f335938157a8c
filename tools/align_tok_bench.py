"""Kernel-level bench of the fused align projection + token-major criterion (csrc/align_tok.hip) at BASELINE config-5 shapes, next to the unfused
path it replaces (library / in-tree GEMM writing the projected feature + csrc/cgd_tok.hip).  Run on the GPU box.

usage: python tools/align_tok_bench.py [--B 8] [--stages 16384,4096,1024,256] [--K 256] [--C 768] [--n 30]
Device time per call by HIP events around back-to-back calls (every call is >= 20 us of kernels)."""
import argparse
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import _lib, ops  # noqa: E402


def t_us(fn, n=30, warm=5):
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
    return 1e3 * ts[len(ts) // 2], 1e3 * ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=8)
    ap.add_argument('--stages', default='16384,4096,1024,256')
    ap.add_argument('--K', type=int, default=256)
    ap.add_argument('--C', type=int, default=768)
    ap.add_argument('--g', type=int, default=8)
    ap.add_argument('--n', type=int, default=30)
    ap.add_argument('--only', default='', help='fwd | bwd | plain: run just that launch at stage 1 (for the profiler)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(1234)
    stages = [int(v) for v in a.stages.split(',')]
    B, K, Cc, g = a.B, a.K, a.C, a.g
    xs = [torch.randn(B, P, K, device=dev).to(torch.bfloat16) for P in stages]
    ts = [(2 * torch.randn(B, P, Cc, device=dev)).to(torch.bfloat16) for P in stages]
    ws = [(torch.randn(Cc, K, device=dev) / K ** 0.5).to(torch.bfloat16) for _ in stages]
    bs = [0.1 * torch.randn(Cc, device=dev) for _ in stages]
    rows = B * (-(-Cc // g))

    def jobs_for(idx, bwd):
        n = len(idx)
        jobs = (ops._AlignTokJob * n)()
        keep = []
        for k, i in enumerate(idx):
            P = stages[i]
            wsb = L.sd_align_cgd_tok_workspace_bytes(B, Cc, P)
            wsp = torch.empty(wsb, dtype=torch.uint8, device=dev)
            lse = torch.zeros(rows, 2, device=dev)
            kl = torch.empty(rows, device=dev)
            loss = torch.empty((), device=dev)
            dY = torch.empty(B, P, Cc, dtype=torch.bfloat16, device=dev)
            dbp = torch.empty(L.sd_align_cgd_tok_tiles(B, P), Cc, device=dev)
            j = jobs[k]
            j.X, j.W, j.bias, j.T = xs[i].data_ptr(), ws[i].data_ptr(), bs[i].data_ptr(), ts[i].data_ptr()
            j.row_lse2, j.row_kl, j.loss, j.workspace, j.workspace_bytes = lse.data_ptr(), kl.data_ptr(), loss.data_ptr(), wsp.data_ptr(), wsb
            j.out, j.db_part = dY.data_ptr(), dbp.data_ptr()
            j.P, j.B, j.K, j.C, j.g, j.inv_tau, j.loss_scale, j.coef = P, B, K, Cc, g, 0.25, 3.0 / rows, 3.0 / (rows * 4.0)
            keep += [wsp, lse, kl, loss, dY, dbp]
        return jobs, keep

    def report(name, us, nbytes, flops):
        print(f'{name:58s} {us[0]:9.1f} us (min {us[1]:8.1f})  {nbytes / us[0] / 1e3:8.1f} GB/s  {flops / us[0] / 1e6:8.1f} TFLOP/s')

    def bytes_of(idx, fwd):
        tot = 0
        for i in idx:
            T = B * stages[i]
            tot += T * K * 2 + T * Cc * 2 * (1 if fwd else 2)
        return tot

    def flops_of(idx):
        return sum(2.0 * B * stages[i] * K * Cc for i in idx)

    all_idx = list(range(len(stages)))
    if a.only:
        jf, kf = jobs_for([0], False)
        T0 = B * stages[0]
        y = torch.empty(T0, Cc, dtype=torch.bfloat16, device=dev)
        fns = {'fwd': lambda: L.sd_align_cgd_tok_fwd_multi(C.cast(jf, C.c_void_p), 1, st), 'bwd': lambda: L.sd_align_cgd_tok_bwd_multi(C.cast(jf, C.c_void_p), 1, st),
               'plain': lambda: L.sd_linear_tok_bf16_fwd(xs[0].data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), y.data_ptr(), T0, K, Cc, st)}
        L.sd_align_cgd_tok_fwd_multi(C.cast(jf, C.c_void_p), 1, st)
        for _ in range(a.n):
            _lib.check(fns[a.only](), a.only)
        torch.cuda.synchronize()
        return
    for idx, tag in [(all_idx, 'all stages'), ([0], f'stage 1 ({B * stages[0]} tokens)')]:
        jf, kf = jobs_for(idx, False)
        # the forward writes row_lse2, which the backward reads: run one forward first so that the backward sees real row constants
        _lib.check(L.sd_align_cgd_tok_fwd_multi(C.cast(jf, C.c_void_p), len(idx), st), 'fwd')
        report(f'fused fwd  (scan + finish), {tag}', t_us(lambda: _lib.check(L.sd_align_cgd_tok_fwd_multi(C.cast(jf, C.c_void_p), len(idx), st), 'fwd'), a.n),
               bytes_of(idx, True), flops_of(idx))
        report(f'fused bwd  (recompute + dY + db), {tag}', t_us(lambda: _lib.check(L.sd_align_cgd_tok_bwd_multi(C.cast(jf, C.c_void_p), len(idx), st), 'bwd'), a.n),
               bytes_of(idx, False), flops_of(idx))
    # stand-alone projection (plain mode) and the library's, stage 1
    T0 = B * stages[0]
    x2, y = xs[0].view(T0, K), torch.empty(T0, Cc, dtype=torch.bfloat16, device=dev)
    gb = T0 * K * 2 + T0 * Cc * 2
    report('plain projection Y = X.W^T + b (ours), stage 1', t_us(lambda: _lib.check(L.sd_linear_tok_bf16_fwd(x2.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), y.data_ptr(), T0, K, Cc, st), 'plain'), a.n),
           gb, 2.0 * T0 * K * Cc)
    b16 = bs[0].to(torch.bfloat16)
    report('F.linear bf16 (library), stage 1', t_us(lambda: F.linear(x2, ws[0], b16), a.n), gb, 2.0 * T0 * K * Cc)
    dy2 = torch.randn(T0, Cc, device=dev).to(torch.bfloat16)
    report('dX = dY @ W (library), stage 1', t_us(lambda: dy2 @ ws[0], a.n), gb, 2.0 * T0 * K * Cc)
    if hasattr(L, 'sd_linear_tok_bf16_bwd_data'):
        dx = torch.empty(T0, K, dtype=torch.bfloat16, device=dev)
        report('dX = dY . W (ours, csrc/align_tok.hip), stage 1',
               t_us(lambda: _lib.check(L.sd_linear_tok_bf16_bwd_data(dy2.data_ptr(), ws[0].data_ptr(), dx.data_ptr(), T0, Cc, K, st), 'bwd_data'), a.n), gb, 2.0 * T0 * K * Cc)
    # the unfused criterion on a stored feature
    yb = y.view(B, stages[0], Cc)
    s_req = yb.clone().requires_grad_(True)

    def unfused_fwd():
        return ops.cgd_kl_tokens(s_req, ts[0], group_size=g, tau=4.0, alpha=3.0)
    report('cgd_tok fwd on the stored feature, stage 1', t_us(unfused_fwd, a.n), 2 * T0 * Cc * 2, 0.0)
    loss = unfused_fwd()

    def unfused_bwd():
        s_req.grad = None
        loss.backward(retain_graph=True)
    report('cgd_tok bwd on the stored feature, stage 1', t_us(unfused_bwd, a.n), 3 * T0 * Cc * 2, 0.0)


if __name__ == '__main__':
    main()

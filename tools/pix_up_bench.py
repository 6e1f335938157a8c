"""PDLoss at the config-2 taps ([8,150,128,128] -> 512 x 512): the fused-upsample pixel-wise kernels (csrc/pix_up.hip) next to the unfused path they
replace (resize.hip x 2 + pix_kl.hip + the resize backward).  python tools/pix_up_bench.py [--dtype f32|bf16]   (run on the GPU box)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import _lib  # noqa: E402
from segdistill_amd.distillation import PDLoss  # noqa: E402


def t_us(fn, n=20, warm=3):
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
    return 1e3 * ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f32')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    torch.manual_seed(0)
    B, C, h, w, H, W = 8, 150, 128, 128, 512, 512
    s = (2 * torch.randn(B, C, h, w, device=dev)).to(dt)
    t = (2 * torch.randn(B, C, h, w, device=dev)).to(dt)
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    DT = 0 if dt == torch.float32 else 1
    wsb = L.sd_pix_kl_up_workspace_bytes(B, h)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    lse2 = torch.empty(2, B * H * W, device=dev)
    loss = torch.empty((), device=dev)
    ds = torch.empty_like(s)
    up = torch.ones((), device=dev)
    rows = B * H * W
    f = lambda: _lib.check(L.sd_pix_kl_up_fwd(s.data_ptr(), t.data_ptr(), DT, B, C, h, w, H, W, 1.0, 1.0 / rows, lse2.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, st), 'fwd')
    b = lambda: _lib.check(L.sd_pix_kl_up_bwd(s.data_ptr(), t.data_ptr(), DT, B, C, h, w, H, W, 1.0, 1.0 / rows, lse2.data_ptr(), up.data_ptr(), ds.data_ptr(), st), 'bwd')
    tf, tb = t_us(f), t_us(b)
    n = B * C * H * W
    print(f'fused PD  fwd {tf:8.1f} us   bwd {tb:8.1f} us   ({a.dtype}; {n / 1e6:.0f} M interpolated (s, t) pairs each way; taps {2 * s.numel() * s.element_size() / 1e6:.0f} MB)')
    gt = torch.zeros(B, 1, H, W, device=dev)
    crit = PDLoss()
    crit.fuse_resize = False
    sg = s.clone().requires_grad_(True)

    def unfused():
        sg.grad = None
        crit(sg, t, gt, 1).backward()
    crit2 = PDLoss()

    def fused_mod():
        sg.grad = None
        crit2(sg, t, gt, 1).backward()
    print(f'module, fused   fwd + bwd {t_us(fused_mod):8.1f} us')
    print(f'module, unfused fwd + bwd {t_us(unfused):8.1f} us   (two resizes, pix_kl, resize backward)')


if __name__ == '__main__':
    main()

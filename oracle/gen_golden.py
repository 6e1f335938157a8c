"""Generate tests/golden/kd_losses.npz by RUNNING THE REFERENCE's own
mmseg/models/distillation/losses.py (imported from /root/reference).

Run in the build container only:   python -m oracle.gen_golden
The output is data (inputs' recipe + the reference's outputs); no reference
source is written anywhere.  TEST INFRASTRUCTURE.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import refimport
from .inputs import kat_pair, wavy_pair, probe_vector

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def run(criterion, s, t, gt_hw, n_iter, dtype=torch.float64):
    """Call a reference criterion exactly as opts.py:103 does; return loss, grad_s."""
    xs = torch.tensor(np.asarray(s), dtype=dtype, requires_grad=True)
    xt = torch.tensor(np.asarray(t), dtype=dtype)
    gt = torch.zeros(xs.shape[0], 1, *gt_hw, dtype=torch.long)
    loss = criterion(xs, xt, gt, n_iter)
    (g,) = torch.autograd.grad(loss, xs)
    return float(loss), g.numpy().astype(np.float64), float(getattr(criterion, 'alpha', float('nan')))


def main():
    L = refimport.load_losses()
    out = {}

    def put(case, **kw):
        for k, v in kw.items():
            out[f'{case}/{k}'] = np.asarray(v)

    # ---- G1: RNG-free known-answer tests, [2,6,4,4] -> 8x8, n_iter=1
    s, t = kat_pair()
    g1 = {
        'cgd_g4_a3_t4': lambda: L.CGDLoss(4, 3, 4),
        'cgd_g3_a3_t2': lambda: L.CGDLoss(3, 3, 2),
        'cgd_g6_a1_t1': lambda: L.CGDLoss(6, 1, 1),
        'cd': lambda: L.CDLoss(),
        'pd': lambda: L.PDLoss(),
        'kld_g2_a2_t3_noresize': lambda: L.KLDLoss(alpha=2, tau=3, transform_config={'loss_type': 'channel', 'group_size': 2}),
        'at': lambda: L.ATLoss(),
    }
    for name, mk in g1.items():
        loss, grad, _ = run(mk(), s, t, (8, 8), 1)
        put(f'G1/{name}', loss=loss, grad=grad)
    # schedules: CGDLossWS at several iterations on ONE instance (alpha is stateful)
    ws = L.CGDLossWS()
    its = [1, 500, 1999, 2000, 2001, 50000, 110000, 110001, 115000, 119999, 120000, 130000]
    its = [i for i in its if i % 1000 != 0 or i in (2000, 110000, 120000, 130000)]
    ws_loss, ws_alpha, ws_it = [], [], []
    for it in its:
        if it % 1000 == 0:
            # shuffle iterations draw torch.randperm from the global CPU RNG: with C=6,g=10
            # all channels are in one group, so the permutation cannot change the loss.
            pass
        loss, _, alpha = run(ws, s, t, (8, 8), it)
        ws_loss.append(loss), ws_alpha.append(alpha), ws_it.append(it)
    put('G1/cgdws_trace', iters=ws_it, loss=ws_loss, alpha=ws_alpha)
    # generic warmup/earlydecay modes on KLDLoss
    for wm in ('linear', 'exp', 'jump'):
        for dm in ('linear', 'exp', 'jump'):
            crit = L.KLDLoss(alpha=2.5, tau=2, resize_config={'mode': 'bilinear', 'align_corners': False},
                             transform_config={'loss_type': 'channel', 'group_size': 3},
                             warmup_config={'mode': wm, 'warmup_iters': 10},
                             earlydecay_config={'mode': dm, 'earlydecay_start': 20, 'earlydecay_end': 30})
            tr_it = [1, 5, 9, 10, 11, 19, 20, 21, 25, 29, 30, 31]
            tr = [run(crit, s, t, (8, 8), it) for it in tr_it]
            put(f'G1/sched_{wm}_{dm}', iters=tr_it, loss=[x[0] for x in tr], alpha=[x[2] for x in tr])
    # IFVD needs labels: deterministic label map
    lab = (np.add.outer(np.arange(8), 2 * np.arange(8)) % 6)[None, None].repeat(2, 0)
    xs = torch.tensor(s, dtype=torch.float64, requires_grad=True)
    loss = L.IFVDLoss()(xs, torch.tensor(t), torch.tensor(lab), 1)
    (g,) = torch.autograd.grad(loss, xs)
    put('G1/ifvd', loss=float(loss), grad=g.numpy(), label=lab)

    # ---- G2: [2,22,16,16] -> 64x64 fp64, pad / no-pad / g==C / g==1 / g>C
    s, t = wavy_pair((2, 22, 16, 16))
    put('G2/inputs', s=s, t=t)
    for (g, a, tau) in [(8, 3, 4), (10, 3, 2), (1, 1, 1), (22, 3, 2), (7, 2, 3), (11, 1.5, 0.5), (32, 1, 2)]:
        loss, grad, _ = run(L.CGDLoss(g, a, tau), s, t, (64, 64), 1)
        put(f'G2/g{g}_a{a}_t{tau}', loss=loss, grad=grad, cfg=[g, a, tau])
    loss, grad, _ = run(L.PDLoss(), s, t, (64, 64), 1)
    put('G2/pd', loss=loss, grad=grad)
    # non-integer / anisotropic resize factors
    for hw in [(40, 56), (16, 16), (24, 100)]:
        loss, grad, _ = run(L.CGDLoss(8, 3, 4), s, t, hw, 1)
        put(f'G2/resize_{hw[0]}x{hw[1]}', loss=loss, grad=grad, hw=hw)

    # ---- G3: shuffle iteration (n_iter % 1000 == 0) with the drawn permutation recorded
    for seed in (0, 7):
        torch.manual_seed(seed)
        perm = torch.randperm(22).numpy()
        torch.manual_seed(seed)
        loss, grad, _ = run(L.CGDLoss(8, 3, 4), s, t, (64, 64), 1000)
        put(f'G3/seed{seed}', loss=loss, grad=grad, perm=perm)

    # ---- G4: ADE-shaped fp32, closed-form inputs (not stored): [2,150,32,32]->128^2 and [1,150,128,128]->512^2
    for tag, shp, hw, cfgs in [
        ('ade32', (2, 150, 32, 32), (128, 128), [(8, 3, 4), (10, 3, 2), (1, 1, 1), (150, 3, 2)]),
        ('ade128', (1, 150, 128, 128), (512, 512), [(8, 3, 4), (1, 1, 1)]),
    ]:
        s4, t4 = wavy_pair(shp)
        probe = probe_vector(shp)
        for (g, a, tau) in cfgs:
            loss64, grad64, _ = run(L.CGDLoss(g, a, tau), s4, t4, hw, 1, torch.float64)
            loss32, grad32, _ = run(L.CGDLoss(g, a, tau), s4, t4, hw, 1, torch.float32)
            put(f'G4/{tag}_g{g}_a{a}_t{tau}', shape=shp, hw=hw, cfg=[g, a, tau], loss64=loss64, loss32=loss32,
                grad_abs_sum=np.abs(grad64).sum(), grad_l2=np.sqrt((grad64 ** 2).sum()),
                grad_probe=(grad64 * probe).sum(), grad_sample=grad64.reshape(-1)[::4099][:64],
                grad32_rel_l2=np.sqrt(((grad32 - grad64) ** 2).sum() / (grad64 ** 2).sum()))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'kd_losses.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'arrays')
    for k in sorted(out):
        if k.endswith('/loss'):
            print(f'  {k:40s} {np.asarray(out[k]).ravel()[:3]}')


if __name__ == '__main__':
    sys.exit(main())

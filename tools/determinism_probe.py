"""Run-to-run determinism of the KD train step: two trainers built from the same seed take the same batches; are their students bit-identical
afterwards?  python tools/determinism_probe.py [--deterministic] [--config configs/kd/cfg2...] [--steps 5] [--graph full|off] [--size 512] [--batch 2]
(reference launches with --deterministic: tools/dist_train.sh:8)."""
import argparse
import copy
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default=os.path.join(ROOT, 'configs', 'kd', 'cfg2_segformer_b2_b0_cgd.py'))
    ap.add_argument('--deterministic', action='store_true')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--graph', default='full')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--batch', type=int, default=2)
    a = ap.parse_args()
    import bench
    from segdistill_amd.config import Config
    from segdistill_amd.engine import KDTrainer, SyntheticADE, set_deterministic
    if a.deterministic:
        set_deterministic(True)
    dev = torch.device('cuda:0')
    cfg = Config.fromfile(a.config)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.manual_seed(0)
        m1 = bench.build_model(cfg, dev)
    m2 = copy.deepcopy(m1)
    # a throw-away trainer first: the FIRST encounter of every convolution / GEMM shape in a process runs the libraries' searches and fills their
    # caches; the two compared runs below then see the same cached choices (what two separate `tools/train.py --deterministic` processes on one
    # machine see after the first has populated MIOpen's user database)
    m0 = copy.deepcopy(m1)
    tr0 = KDTrainer(m0, dict(cfg.optimizer), dict(cfg.lr_config), world=1, precision=cfg.get('precision'))
    d0 = SyntheticADE(a.batch, size=(a.size, a.size), device=dev, pool=3, seed=1)
    if a.graph == 'full':
        tr0.enable_graph(dict(img=d0._pool[0][0], img_metas=None, gt_semantic_seg=d0._pool[0][1]))
    for _ in range(2):
        tr0.step(d0.next())
    torch.cuda.synchronize()
    del tr0, m0, d0
    res = []
    for m in (m1, m2):
        tr = KDTrainer(m, dict(cfg.optimizer), dict(cfg.lr_config), world=1, precision=cfg.get('precision'))
        data = SyntheticADE(a.batch, size=(a.size, a.size), device=dev, pool=3, seed=1)
        if a.graph == 'full':
            b0 = data._pool[0]
            ok = tr.enable_graph(dict(img=b0[0], img_metas=None, gt_semantic_seg=b0[1]))
            print('graph:', ok, getattr(tr, 'graph_error', None))
        torch.manual_seed(7)
        for _ in range(a.steps):
            tr.step(data.next())
        torch.cuda.synchronize()
        res.append({n: p.detach().clone() for n, p in m.student.named_parameters()})
        del tr
    bad = [(n, float((res[0][n] - res[1][n]).abs().max())) for n in res[0] if not torch.equal(res[0][n], res[1][n])]
    print(f'deterministic={a.deterministic} steps={a.steps}: {len(bad)} of {len(res[0])} student tensors differ between the two runs')
    for n, d in bad[:12]:
        print(f'   {n:60s} max |diff| {d:.3e}')
    sys.exit(1 if bad and a.deterministic else 0)


if __name__ == '__main__':
    main()

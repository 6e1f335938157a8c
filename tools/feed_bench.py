"""Throughput of the real-data feed (reference local_configs/_base_/datasets/ade20k_repeat.py:7-18): the ADE20K training pipeline
of configs/kd/cfg2_segformer_b2_b0_cgd_ade20k.py over a SYNTHETIC image tree (JPEG 683 x 512 + PNG label maps, ADE20K's median size; no dataset offline).

    python tools/feed_bench.py [--images 400] [--workers 1,4,8,16] [--batch 8] [--batches 40] [--root /tmp/fake_ade]

Prints (1) the cost of every pipeline stage on one core, (2) DataLoader imgs/s per worker count (CPU side only: what the host can deliver),
and with --gpu-step-ms T the GPU idle fraction such a feed would leave at a step of T ms per batch (config 2: ~10.4 ms for 8 images).
On the GPU box the same loader can be put in front of the real step: tools/train.py <config> --data-root ROOT --iters N."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fake_ade(root, n, size=(512, 683), seed=0):
    """ADE20K-shaped tree: smooth random images (JPEG decode cost depends on content: white noise is ~2x slower than photographs), label maps of
    a few rectangles out of 151 classes."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, 'images/training'), exist_ok=True)
    os.makedirs(os.path.join(root, 'annotations/training'), exist_ok=True)
    h, w = size
    for i in range(n):
        f_img = os.path.join(root, 'images/training', f'ADE_train_{i:08d}.jpg')
        if os.path.isfile(f_img):
            continue
        low = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(low).resize((w, h), Image.BICUBIC))
        img = np.clip(img.astype(np.int16) + rng.integers(-12, 13, img.shape, dtype=np.int16), 0, 255).astype(np.uint8)
        seg = np.zeros((h, w), np.uint8)
        for _ in range(12):
            y, x = int(rng.integers(0, h - 32)), int(rng.integers(0, w - 32))
            seg[y:y + int(rng.integers(32, h // 2)), x:x + int(rng.integers(32, w // 2))] = int(rng.integers(1, 151))
        Image.fromarray(img).save(f_img, quality=90)
        Image.fromarray(seg).save(os.path.join(root, 'annotations/training', f'ADE_train_{i:08d}.png'))
    return root


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=400)
    ap.add_argument('--workers', default='1,4,8,16')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--batches', type=int, default=40)
    ap.add_argument('--root', default='/tmp/fake_ade')
    ap.add_argument('--gpu-step-ms', type=float, default=10.4)
    a = ap.parse_args()
    import torch
    import segdistill_amd
    from segdistill_amd.config import Config
    from segdistill_amd.data import build_dataloader, build_dataset
    from segdistill_amd.data.pipelines import PIPELINES
    from segdistill_amd.registry import build_from_cfg
    segdistill_amd.register_all()
    root = fake_ade(a.root, a.images)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'kd', 'cfg2_segformer_b2_b0_cgd_ade20k.py'))
    train = cfg.data.train.to_dict() if hasattr(cfg.data.train, 'to_dict') else dict(cfg.data.train)
    train['dataset']['data_root'] = root
    ds = build_dataset(train)
    print(f'{len(ds)} samples ({a.images} images x RepeatDataset), pipeline of configs/kd/cfg2_segformer_b2_b0_cgd_ade20k.py, host cores: {os.cpu_count()}')

    # (1) per-stage cost on one core
    inner = ds.dataset if hasattr(ds, 'dataset') else ds
    from segdistill_amd.data.pipelines import Compose
    stages = Compose([dict(c) for c in train['dataset']['pipeline']]).transforms          # Compose links Resize -> RandomCrop (crop-aware resize)
    acc = [0.0] * len(stages)
    n_probe = min(60, a.images)
    for i in range(n_probe):
        results = dict(img_info=inner.img_infos[i], ann_info=inner.get_ann_info(i))
        inner.pre_pipeline(results)
        for k, st in enumerate(stages):
            t0 = time.perf_counter()
            results = st(results)
            acc[k] += time.perf_counter() - t0
    tot = sum(acc)
    print(f'one core, mean over {n_probe} images: {tot / n_probe * 1e3:.1f} ms per image = {n_probe / tot:.1f} imgs/s per worker')
    for st, v in zip(stages, acc):
        print(f'   {type(st).__name__:24s} {v / n_probe * 1e3:7.2f} ms  {100 * v / tot:5.1f} %')

    # (2) loader throughput
    for nw in [int(v) for v in a.workers.split(',')]:
        loader = build_dataloader(ds, a.batch, nw, world=1, rank=0, shuffle=True, seed=0, pin_memory=False)
        it = iter(loader)
        for _ in range(max(2, nw // 2)):          # workers start up, first batches
            next(it)
        t0 = time.perf_counter()
        for _ in range(a.batches):
            b = next(it)
        dt = time.perf_counter() - t0
        rate = a.batches * a.batch / dt
        need = a.batch / (a.gpu_step_ms * 1e-3)
        idle = max(0.0, 1.0 - rate / need)
        print(f'workers {nw:3d}: {rate:8.1f} imgs/s  (batch {tuple(b["img"].shape)}; a {a.gpu_step_ms} ms step wants {need:.0f} imgs/s -> GPU idle {100 * idle:.0f} %)')
        del it, loader


if __name__ == '__main__':
    main()

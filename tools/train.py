#!/usr/bin/env python3
"""KD training driver (the role of the reference's tools/train.py + mmseg/apis/train.py for the SDModule path).

    python tools/train.py CONFIG [--iters N] [--work-dir DIR] [--resume-from CKPT] [--seed S] [--graph auto|on|hybrid|off]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py CONFIG --launcher pytorch

CONFIG is a reference-dialect config file (the reference's local_configs/** KD configs load unchanged, or configs/kd/*.py).
Data: the config's data.train (the reference's ADE20K dataset + pipeline dialect, segdistill_amd/data -- DESIGN.md section 6c; `--data-root`
points it at a directory tree) or, when the config has no data.train entry, synthetic ADE20K-shaped batches.
Per iteration, like mmcv's IterBasedRunner + OptimizerHook: lr update -> zero grad -> train_step -> backward ->
gradient all-reduce -> optimizer step; a text log line every log_config.interval iterations; a checkpoint every
checkpoint_config.interval iterations (student + optimizer + iteration + distillation step counter)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config')
    ap.add_argument('--work-dir', default=None)
    ap.add_argument('--iters', type=int, default=None, help='stop after this many iterations (default: runner.max_iters)')
    ap.add_argument('--resume-from', default=None)
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none')
    ap.add_argument('--graph', choices=['auto', 'on', 'hybrid', 'off'], default='auto')
    ap.add_argument('--synthetic-weights', action='store_true', help='train from random init when checkpoints named by the config are absent')
    ap.add_argument('--data-root', default=None, help='dataset directory for configs with data.train (overrides its data_root); '
                                                      'without a data.train entry synthetic ADE20K-shaped batches are used')
    ap.add_argument('--deterministic', action='store_true', help='run-to-run bit-identical training (reference tools/dist_train.sh:8): MIOpen held to its '
                    'deterministic convolution algorithms, auto-tuner off; the HIP kernels of this package are deterministic by construction')
    ap.add_argument('--cpu-plumbing', action='store_true', help='BASELINE configs[0]: run the harness on a box WITHOUT a GPU -- the KLDLoss criteria evaluate '
                    'CPU taps with ATen ops (never used for CUDA tensors; without it a CPU tap raises)')
    ap.add_argument('--options', nargs='*', default=[], help='config overrides key=value (dotted keys)')
    args = ap.parse_args()

    import segdistill_amd
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.config import Config
    from segdistill_amd.engine import KDTrainer, SyntheticADE, init_distributed
    from segdistill_amd.segmentors import sd_module
    rank, local, world = init_distributed() if args.launcher == 'pytorch' or int(os.environ.get('WORLD_SIZE', '1')) > 1 else (0, 0, 1)
    device = torch.device('cuda', local) if torch.cuda.is_available() else torch.device('cpu')
    if device.type == 'cuda':
        torch.cuda.set_device(device)
        torch.backends.cudnn.benchmark = True
    if args.deterministic:
        from segdistill_amd.engine import set_deterministic
        set_deterministic(True)
    if args.cpu_plumbing:
        if device.type == 'cuda':
            raise SystemExit('--cpu-plumbing is for boxes without a GPU; on a GPU the HIP kernels are the only path')
        from segdistill_amd.distillation import losses as kd_losses
        kd_losses.CPU_PLUMBING = True
    elif device.type == 'cpu':
        print('no GPU visible: the distillation criteria are HIP kernels and will raise on CPU taps (pass --cpu-plumbing to exercise the harness '
              'with ATen criteria: BASELINE configs[0])', file=sys.stderr)
    cfg = Config.fromfile(args.config)
    if args.options:
        import ast
        over = {}
        for kv in args.options:
            k, v = kv.split('=', 1)
            try:
                over[k] = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                over[k] = v
        cfg.merge_from_dict(over)
    segdistill_amd.register_all()
    sd_module.SYNTHETIC_WEIGHTS_OK = args.synthetic_weights
    torch.manual_seed(args.seed)
    model = build_segmentor(dict(cfg.model)).to(device)
    max_iters = int(cfg.runner.max_iters) if 'runner' in cfg else 160000
    n_iters = args.iters or max_iters
    log_every = int(cfg.get('log_config', {}).get('interval', 50))
    ckpt_every = int(cfg.get('checkpoint_config', {}).get('interval', 4000))
    trainer = KDTrainer(model, dict(cfg.optimizer), dict(cfg.lr_config), max_iters=max_iters, world=world, log_interval=log_every,
                        precision=cfg.get('precision'))
    if args.resume_from:
        trainer.resume(args.resume_from, map_location=device)
        if rank == 0:
            print(f'resumed from {args.resume_from}: iter {trainer.iter}, distillation step {model.cnt}')
    B = int(cfg.data.samples_per_gpu)
    if cfg.data.get('train') is not None:
        # real data: the reference's dataset + pipeline configs (segdistill_amd/data), one loader per rank
        from segdistill_amd.data import build_dataloader, build_dataset
        from segdistill_amd.data.feed import LoaderFeed
        tcfg = cfg.data.train.to_dict() if hasattr(cfg.data.train, 'to_dict') else dict(cfg.data.train)
        inner = tcfg['dataset'] if tcfg.get('type') == 'RepeatDataset' else tcfg
        if args.data_root:
            inner['data_root'] = args.data_root
        dataset = build_dataset(tcfg)
        loader = build_dataloader(dataset, B, int(cfg.data.get('workers_per_gpu', 2)), world=world, rank=rank, seed=args.seed)
        data = LoaderFeed(loader, device)
        if rank == 0:
            print(f'dataset: {len(dataset)} samples ({type(dataset).__name__}), {B} per GPU, {world} rank(s)')
    else:
        data = SyntheticADE(B, size=tuple(cfg.get('crop_size', (512, 512))), num_classes=int(cfg.get('num_classes', 150)), seed=args.seed,
                            rank=rank, device=device)
    work = args.work_dir or os.path.join(ROOT, 'work_dirs', os.path.splitext(os.path.basename(args.config))[0])
    if rank == 0:
        os.makedirs(work, exist_ok=True)
    mode = args.graph if args.graph != 'auto' else 'on'
    warm = 0
    t0 = time.perf_counter()
    cur = data.next()
    while trainer.iter < n_iters:
        if warm == 3 and device.type == 'cuda' and mode != 'off':
            # the batch about to be trained on serves as the capture's example (no batch is consumed by the capture; the student's
            # BatchNorm buffers are restored after its warm-up passes -- KDTrainer.enable_graph)
            if not (mode == 'on' and trainer.enable_graph(cur)):
                trainer.enable_hybrid_graph(cur)
        nxt = data.next() if trainer.iter + 1 < n_iters else None
        trainer.step(cur, nxt)                # the frozen teacher's forward for `nxt` overlaps this iteration's backward
        cur = nxt
        warm += 1
        it = trainer.iter
        if it % log_every == 0 or it == n_iters:
            vals = trainer.log_values()  # the only device->host sync
            if rank == 0:
                dt = (time.perf_counter() - t0) / max(1, log_every)
                t0 = time.perf_counter()
                lr = trainer.optimizer.param_groups[0]['lr']
                print(f'Iter [{it}/{n_iters}]\tlr: {lr:.3e}, time: {dt:.3f}, ' + ', '.join(f'{k}: {v:.4f}' for k, v in vals.items()), flush=True)
        if rank == 0 and (it % ckpt_every == 0 or it == n_iters):
            trainer.save(os.path.join(work, 'latest.pth'))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

"""Worker of tests/test_segmented_graph_gpu.py: one rank of a (possibly one-rank) process group on cuda:0.  Trains the same
tiny SegFormer KD model twice from identical weights -- eagerly with torch's SyncBatchNorm path, and through the SEGMENTED
hipGraph step (engine/segments.py) -- and prints one JSON line with what the test compares."""
import copy
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def pspnet_main():
    """A ResNet student (SyncBN throughout) with ranks > 1: the segmented capture must decline BEFORE touching the device state, the
    hybrid mode must leave the backbone eager (its collectives cannot be recorded) and still graph the teacher, and training goes on."""
    import segdistill_amd
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.engine import KDTrainer, SyntheticADE, init_distributed
    from segdistill_amd.segmentors import sd_module
    rank, local, world = init_distributed()
    torch.cuda.set_device(0)
    segdistill_amd.register_all()
    norm = dict(type='SyncBN', requires_grad=True)

    def seg(depth):
        ch = 512 if depth == 18 else 2048
        return dict(type='EncoderDecoder', pretrained=None,
                    backbone=dict(type='ResNetV1c', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 2, 4), strides=(1, 2, 1, 1),
                                  norm_cfg=norm, norm_eval=False, style='pytorch', contract_dilation=True),
                    decode_head=dict(type='PSPHead', in_channels=ch, in_index=3, channels=64, pool_scales=(1, 2, 3, 6), dropout_ratio=0.1,
                                     num_classes=150, norm_cfg=norm, align_corners=False,
                                     loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    cfg = dict(type='SDModule', cfg_s=seg(18), cfg_t=seg(18),
               distillation=[dict(student_layer='decode_head.conv_seg', teacher_layer='decode_head.conv_seg', loss_name='CDLoss', loss_config={})],
               t_pretrain=None, train_cfg=dict(), test_cfg=dict(mode='whole'))
    sd_module.SYNTHETIC_WEIGHTS_OK = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.manual_seed(0)
        model = build_segmentor(cfg).cuda()
    tr = KDTrainer(model, dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01), None, world=world)
    data = SyntheticADE(2, size=(128, 128), device='cuda:0', pool=2, seed=1, rank=rank)
    out = {'rank': rank, 'world': world, 'mode': 'pspnet'}
    tr.step(data.next())
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        out['full'] = bool(tr.enable_graph(data.next()))
        out['segments'] = len(tr._seg.items) if out['full'] else 0
        out['hybrid'] = False if out['full'] else bool(tr.enable_hybrid_graph(data.next()))
    out['warnings'] = [str(w.message)[:200] for w in caught if 'capture failed' in str(w.message)]
    out['backbone_graphed'] = getattr(model.student, '_graphed_backbone', None) is not None
    out['teacher_graphed'] = model._graphed_teacher is not None
    vals = []
    cur = data.next()
    for _ in range(3):
        nxt = data.next()
        tr.step(cur, nxt)
        cur = nxt
        vals.append(tr.log_values())
    out['steps'] = vals
    out['digest'] = float(sum(p.detach().double().sum() for p in model.student.parameters()))
    sys.stdout.write('RESULT ' + json.dumps(out) + '\n')     # one write call per record (two ranks share the pipe)
    sys.stdout.flush()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def main(upstream_tap=False):
    import segdistill_amd
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.engine import KDTrainer, SyntheticADE, init_distributed
    from segdistill_amd.layers import ChainedSyncBatchNorm
    from segdistill_amd.segmentors import sd_module
    rank, local, world = init_distributed()
    torch.cuda.set_device(0)
    segdistill_amd.register_all()
    norm = dict(type='SyncBN', requires_grad=True)

    def seg(v, ch, e):
        return dict(type='EncoderDecoder', pretrained=None, backbone=dict(type=f'mit_{v}', style='pytorch'),
                    decode_head=dict(type='SegFormerHead', in_channels=ch, in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32], channels=128,
                                     dropout_ratio=1e-12, num_classes=150, norm_cfg=norm, align_corners=False, decoder_params=dict(embed_dim=e),
                                     loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    bil = dict(mode='bilinear', align_corners=False)
    distillation = [dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='KLDLoss',
                         loss_config=dict(alpha=3, tau=4, resize_config=bil, shuffle_config={'interval': 3},
                                          transform_config={'loss_type': 'channel', 'group_size': 8}))]
    if upstream_tap:
        # a KD tap IN FRONT of the chained norm (config 5 taps decode_head.linear_c1..4): backbone and linear_c1 parameters get a KD
        # gradient in the walk from the loss and a CE gradient in the walk that continues behind the norm's backward
        distillation.append(dict(student_layer='decode_head.linear_c1', teacher_layer='decode_head.linear_c1', loss_name='KLDLoss',
                                 loss_config=dict(alpha=30, tau=1, transform_config={'loss_type': 'channel', 'group_size': 8})))
    cfg = dict(type='SDModule', cfg_s=seg('b0', [32, 64, 160, 256], 256), cfg_t=seg('b0', [32, 64, 160, 256], 256),
               distillation=distillation, t_pretrain=None, train_cfg=dict(), test_cfg=dict(mode='whole'))
    sd_module.SYNTHETIC_WEIGHTS_OK = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        torch.manual_seed(0)
        ref = build_segmentor(cfg)
    ref.student.backbone.reset_drop_path(0.)
    if upstream_tap:
        with torch.no_grad():   # identical student / teacher initialisation would make the feature-level KD term (and its gradient) vanish
            for p in ref.teacher.parameters():
                p.mul_(1.25)
    ref = ref.cuda()
    gra = copy.deepcopy(ref)
    bn_e, bn_g = ref.student.decode_head.linear_fuse.norm, gra.student.decode_head.linear_fuse.norm
    out = {'rank': rank, 'world': world, 'chained': isinstance(bn_g, ChainedSyncBatchNorm)}
    opt = dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    t_e = KDTrainer(ref, opt, dict(policy='poly', power=1.0, min_lr=0.0, by_epoch=False), world=world)
    t_g = KDTrainer(gra, opt, dict(policy='poly', power=1.0, min_lr=0.0, by_epoch=False), world=world)
    data_e = SyntheticADE(2, size=(128, 128), device='cuda:0', pool=3, seed=1, rank=rank)
    data_g = SyntheticADE(2, size=(128, 128), device='cuda:0', pool=3, seed=1, rank=rank)
    example = dict(img=data_g._pool[0][0], img_metas=None, gt_semantic_seg=data_g._pool[0][1])
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        ok = t_g.enable_graph(example)
    out['graph'] = bool(ok)
    out['warnings'] = [str(w.message)[:300] for w in caught if 'capture failed' in str(w.message)]
    if ok:
        out['graphs'] = len([g for g in t_g._seg.items if isinstance(g, torch.cuda.CUDAGraph)])
        out['cuts'] = t_g._seg.cuts
    out['cnt_after_capture'] = gra.cnt
    out['tracked_after_capture'] = int(bn_g.num_batches_tracked) - int(bn_e.num_batches_tracked)
    out['running_stats_moved_by_capture'] = float((bn_g.running_mean - bn_e.running_mean).abs().max() + (bn_g.running_var - bn_e.running_var).abs().max())
    steps = []
    cur = data_g.next()
    for it in range(5):
        torch.manual_seed(100 + it)
        t_e.step(data_e.next())
        torch.manual_seed(100 + it)
        nxt = data_g.next()
        t_g.step(cur, nxt)
        cur = nxt
        steps.append((t_e.log_values(), t_g.log_values()))
    out['steps'] = steps
    num = sum(float((a - b).pow(2).sum()) for a, b in zip(ref.student.parameters(), gra.student.parameters()))
    den = sum(float(a.pow(2).sum()) for a in ref.student.parameters())
    out['param_rel_l2'] = (num / den) ** 0.5
    out['running_mean_diff'] = float((bn_e.running_mean - bn_g.running_mean).abs().max())
    out['running_var_rel'] = float(((bn_e.running_var - bn_g.running_var).abs() / bn_e.running_var.abs().clamp_min(1e-6)).max())
    out['tracked'] = [int(bn_e.num_batches_tracked), int(bn_g.num_batches_tracked)]
    out['digest'] = float(sum(p.detach().double().sum() for p in gra.student.parameters()))
    sys.stdout.write('RESULT ' + json.dumps(out) + '\n')     # one write call per record (two ranks share the pipe)
    sys.stdout.flush()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else ''
    pspnet_main() if mode == 'pspnet' else main(upstream_tap=(mode == 'upstream'))

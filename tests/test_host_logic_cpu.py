"""Host-side logic on the CPU: registry / config semantics, criterion schedules, loss naming,
the flat-buffer DP reducer under gloo (world_size 2) and the full engine step with the eager
oracle criteria swapped in."""
import os
import sys
import textwrap

import numpy as np
import copy

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_and_build_from_cfg():
    from segdistill_amd.registry import Registry, build_from_cfg
    reg = Registry('thing')

    @reg.register_module()
    class Foo:
        def __init__(self, a, b=2, train_cfg=None):
            self.a, self.b, self.train_cfg = a, b, train_cfg

    obj = build_from_cfg(dict(type='Foo', a=1), reg, default_args=dict(b=5, train_cfg='x'))
    assert (obj.a, obj.b, obj.train_cfg) == (1, 5, 'x')
    obj = build_from_cfg(dict(type='Foo', a=1, b=7), reg, default_args=dict(b=5))
    assert obj.b == 7  # the config wins over default_args
    with pytest.raises(KeyError):
        build_from_cfg(dict(type='Nope'), reg)
    with pytest.raises(KeyError):
        reg.register_module()(Foo)  # duplicate


def test_config_base_delete_and_options(tmp_path):
    from segdistill_amd.config import Config
    (tmp_path / 'base.py').write_text("optimizer = dict(type='SGD', lr=0.01, momentum=0.9)\nmodel = dict(type='A', head=dict(c=1, d=2))\nx = 1\n")
    (tmp_path / 'child.py').write_text(textwrap.dedent('''
        _base_ = ['./base.py']
        t = '3'
        picked = eval(f"dict(n={t})")
        optimizer = dict(_delete_=True, type='AdamW', lr=6e-5)
        model = dict(head=dict(c=10))
    '''))
    cfg = Config.fromfile(str(tmp_path / 'child.py'))
    assert cfg.optimizer == {'type': 'AdamW', 'lr': 6e-5}          # replaced, not merged
    assert cfg.model.head == {'c': 10, 'd': 2} and cfg.model.type == 'A'  # deep merge, child wins
    assert cfg.picked.n == 3 and cfg.x == 1
    cfg.merge_from_dict({'model.head.d': 5, 'data.samples_per_gpu': 8})
    assert cfg.model.head.d == 5 and cfg.data.samples_per_gpu == 8


@pytest.mark.reference
def test_reference_kd_configs_load_unchanged():
    ref = '/root/reference/local_configs'
    if not os.path.isdir(ref):
        pytest.skip('reference tree not present')
    import segdistill_amd
    from segdistill_amd.builder import DISTILL_LOSSES
    from segdistill_amd.config import Config
    segdistill_amd.register_all()
    n = 0
    for sub in ('exp_tab5', 'Teacher_Student_Size', 'Group_Size', 'Weight_Temperature'):
        for f in sorted(os.listdir(os.path.join(ref, sub))):
            if not f.endswith('.py'):
                continue
            cfg = Config.fromfile(os.path.join(ref, sub, f))
            if cfg.model.type != 'SDModule':
                continue
            for d in cfg.model.distillation:
                assert d['loss_name'] in DISTILL_LOSSES, (f, d['loss_name'])
            assert cfg.optimizer.type == 'AdamW' and 'custom_keys' in cfg.optimizer.paramwise_cfg
            n += 1
    assert n >= 20


def test_criterion_schedules_match_golden(golden):
    """KLDLoss.warmup/earlydecay state machine vs the alpha trace recorded from the reference."""
    import segdistill_amd
    from segdistill_amd.distillation import CGDLossWS, KLDLoss
    c = CGDLossWS()
    for it, alpha in zip(golden['G1/cgdws_trace/iters'], golden['G1/cgdws_trace/alpha']):
        c.warmup(int(it))
        c.earlydecay(int(it))
        assert c.alpha == pytest.approx(float(alpha), abs=1e-15), it
    for wm in ('linear', 'exp', 'jump'):
        for dm in ('linear', 'exp', 'jump'):
            c = KLDLoss(alpha=2.5, tau=2, warmup_config={'mode': wm, 'warmup_iters': 10},
                        earlydecay_config={'mode': dm, 'earlydecay_start': 20, 'earlydecay_end': 30})
            key = f'G1/sched_{wm}_{dm}'
            for it, alpha in zip(golden[key + '/iters'], golden[key + '/alpha']):
                c.warmup(int(it))
                c.earlydecay(int(it))
                assert c.alpha == pytest.approx(float(alpha), abs=1e-15), (wm, dm, it)


def test_preset_fields_match_reference_defaults():
    from segdistill_amd.distillation import CDLoss, CGDLoss, CGDLossWS, PDLoss
    c = CGDLoss()
    assert (c.alpha_0, c.tau, c.transform_config['group_size'], c.shuffle_config['interval']) == (3, 2, 10, 1000)
    c = CGDLoss(group_size=8, alpha=3, tau=4)
    assert c.transform_config == {'loss_type': 'channel', 'group_size': 8} and c.tau == 4
    assert CDLoss().transform_config == {'loss_type': 'channel', 'group_size': 1} and CDLoss().shuffle_config is None
    assert PDLoss().transform_config == {'loss_type': 'pixel'}
    w = CGDLossWS()
    assert w.warmup_config == {'mode': 'linear', 'warmup_iters': 2000}
    assert w.earlydecay_config == {'mode': 'linear', 'earlydecay_start': 110000, 'earlydecay_end': 120000}


def _tiny_sd_cfg():
    norm = dict(type='SyncBN', requires_grad=True)
    def seg(v, ch, e):
        return dict(type='EncoderDecoder', pretrained=None, backbone=dict(type=f'mit_{v}', style='pytorch'),
                    decode_head=dict(type='SegFormerHead', in_channels=ch, in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32], channels=128,
                                     dropout_ratio=0.1, num_classes=150, norm_cfg=norm, align_corners=False, decoder_params=dict(embed_dim=e),
                                     loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    return dict(type='SDModule', cfg_s=seg('b0', [32, 64, 160, 256], 256), cfg_t=seg('b0', [32, 64, 160, 256], 256),
                distillation=[dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='CGDLoss',
                                   loss_config={'group_size': 8, 'alpha': 3, 'tau': 4}),
                              dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='CDLoss',
                                   loss_config={})],
                t_pretrain=None, train_cfg=dict(), test_cfg=dict(mode='whole'))


def test_sdmodule_contract_and_engine_step_cpu():
    import segdistill_amd
    from oracle.eager_modules import swap_in_eager_criteria
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.engine import KDTrainer, SyntheticADE
    from segdistill_amd.segmentors import sd_module
    segdistill_amd.register_all()
    with pytest.raises(FileNotFoundError):
        build_segmentor(_tiny_sd_cfg())  # the teacher checkpoint is mandatory unless synthetic-weights mode is on
    sd_module.SYNTHETIC_WEIGHTS_OK = True
    try:
        torch.manual_seed(0)
        model = build_segmentor(_tiny_sd_cfg())
    finally:
        sd_module.SYNTHETIC_WEIGHTS_OK = False
    assert not any(p.requires_grad for p in model.teacher.parameters())
    model.train()
    assert model.student.training and not model.teacher.training  # Q1: the teacher stays in eval
    # the product criteria are HIP-only: on CPU tensors they must raise, not fall back
    data = SyntheticADE(1, size=(64, 64), device='cpu', pool=1)
    with pytest.raises(RuntimeError, match='GPU only'):
        model.train_step(data.next(), None)
    model.cnt = 0
    swap_in_eager_criteria(model)
    opt = dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01,
               paramwise_cfg=dict(custom_keys={'pos_block': dict(decay_mult=0.), 'norm': dict(decay_mult=0.), 'head': dict(lr_mult=10.)}))
    tr = KDTrainer(model, opt, dict(policy='poly', warmup='linear', warmup_iters=1500, warmup_ratio=1e-6, power=1.0, min_lr=0.0, by_epoch=False))
    before = model.student.decode_head.linear_pred.weight.detach().clone()
    out = tr.step(data.next())
    assert set(out) == {'loss', 'log_vars', 'num_samples'} and out['num_samples'] == 1
    keys = list(out['log_vars'])
    assert keys[:2] == ['decode.loss_seg', 'decode.acc_seg'] and keys[-1] == 'loss'
    kd = [k for k in keys if k.startswith('loss_decode_head.linear_pred<->decode_head.linear_pred_')]
    assert len(kd) == 2 and kd[1].endswith('#1')  # Q3: colliding names are kept apart
    vals = tr.log_values()
    assert vals['loss'] == pytest.approx(sum(v for k, v in vals.items() if 'loss' in k and k != 'loss'), rel=1e-5)
    assert model.cnt == 1
    assert not torch.equal(before, model.student.decode_head.linear_pred.weight)  # the step moved the student
    assert model.student.decode_head.conv_seg.weight.requires_grad is False      # Q12: the dead parameter is frozen
    lrs = sorted({g['initial_lr'] for g in tr.optimizer.param_groups})
    assert lrs == pytest.approx([6e-5, 6e-4])
    # poly + linear warm-up at it=0: lr * (1 - (1 - 0) * (1 - 1e-6)) = lr * 1e-6
    assert tr.sched.lr_at(6e-5, 0) == pytest.approx(6e-5 * 1e-6, rel=1e-6)
    assert tr.sched.lr_at(6e-5, 1500) == pytest.approx(6e-5 * (1 - 1500 / 160000), rel=1e-9)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from segdistill_amd.engine import DataParallelReducer, init_distributed
    from segdistill_amd.segmentors.base import parse_losses
    init_distributed(backend='gloo')
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    red = DataParallelReducer(net.parameters(), world=world)
    red.broadcast_parameters(net)
    full = torch.arange(4 * 6, dtype=torch.float32).reshape(4, 6) / 10.0
    x = full[rank * 2:(rank + 1) * 2]  # batch shard
    red.zero_grad()
    loss = net(x).pow(2).mean()
    loss.backward()
    red.all_reduce()
    _, logs = parse_losses({'loss_a': loss, 'acc': loss.detach() * 2})
    # numpy (pickled by value): torch tensors travel through shared-memory handles that die with the child
    q.put((rank, red.flat.numpy().copy(), [p.detach().numpy().copy() for p in net.parameters()], logs))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_buffer_dp_matches_single_process_gloo():
    world, port = 2, 29741
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, p0, l0), (_, g1, p1, l1) = res
    g0, g1 = torch.from_numpy(g0), torch.from_numpy(g1)
    p0, p1 = [torch.from_numpy(a) for a in p0], [torch.from_numpy(a) for a in p1]
    assert torch.equal(g0, g1)                       # every rank holds the same averaged gradient
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)                     # parameters were broadcast from rank 0
    # single-process oracle: same parameters, the concatenated batch
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    with torch.no_grad():
        for p, v in zip(net.parameters(), p0):
            p.copy_(v)
    full = torch.arange(4 * 6, dtype=torch.float32).reshape(4, 6) / 10.0
    loss = net(full).pow(2).mean()
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert torch.allclose(g0, flat, rtol=1e-5, atol=1e-7)   # mean of shard losses == loss of the whole batch (equal shards)
    assert l0['loss'] == pytest.approx(float(loss), rel=1e-5) and l0 == l1
    assert l0['acc'] == pytest.approx(2 * float(loss), rel=1e-5)


def test_cross_entropy_known_answers_from_reference_tests():
    """Known answers held by the reference's own tests/test_models/test_losses.py:55-69 for the softmax
    CrossEntropyLoss (the sigmoid / mask variants :71-84 are outside the KD path and raise here)."""
    import segdistill_amd
    from segdistill_amd.builder import build_loss
    from segdistill_amd.losses import accuracy
    segdistill_amd.register_all()
    pred, label = torch.Tensor([[100, -100]]), torch.Tensor([1]).long()
    crit = build_loss(dict(type='CrossEntropyLoss', use_sigmoid=False, class_weight=[0.8, 0.2], loss_weight=1.0))
    assert torch.allclose(crit(pred, label), torch.tensor(40.))
    crit = build_loss(dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    assert torch.allclose(crit(pred, label), torch.tensor(200.))
    with pytest.raises(NotImplementedError):
        build_loss(dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0))
    # accuracy known answers (reference tests/test_models/test_losses.py:89-144)
    pred = torch.Tensor([[0.2, 0.3, 0.6, 0.5], [0.1, 0.1, 0.2, 0.6], [0.9, 0.0, 0.0, 0.1], [0.4, 0.7, 0.1, 0.1], [0.0, 0.0, 0.99, 0]])
    assert accuracy(pred, torch.Tensor([2, 3, 0, 1, 2]).long()).item() == pytest.approx(100.)
    assert accuracy(pred, torch.Tensor([2, 2, 0, 1, 0]).long()).item() == pytest.approx(60.)  # 3 of 5 top-1 hits
    a1, a2 = accuracy(pred, torch.Tensor([2, 3, 0, 1, 2]).long(), topk=(1, 2))
    assert a1.item() == pytest.approx(100.) and a2.item() == pytest.approx(100.)


def _engine_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import warnings
    import segdistill_amd
    from oracle.eager_modules import swap_in_eager_criteria
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.distillation import CGDLoss
    from segdistill_amd.engine import KDTrainer, SyntheticADE, init_distributed
    from segdistill_amd.segmentors import sd_module
    torch.set_num_threads(2)
    init_distributed(backend='gloo')
    segdistill_amd.register_all()
    sd_module.SYNTHETIC_WEIGHTS_OK = True
    torch.manual_seed(1234 + rank)            # different initial weights per rank: the trainer must broadcast rank 0's
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = build_segmentor(_tiny_sd_cfg())
    swap_in_eager_criteria(model)
    opt = dict(type='AdamW', lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01,
               paramwise_cfg=dict(custom_keys={'norm': dict(decay_mult=0.), 'head': dict(lr_mult=10.)}))
    tr = KDTrainer(model, opt, dict(policy='poly', power=1.0, min_lr=0.0, by_epoch=False), world=world)
    data = SyntheticADE(1, size=(64, 64), device='cpu', pool=2, seed=0, rank=rank)
    logs = []
    for _ in range(2):
        tr.step(data.next())
        logs.append(tr.log_values())
    # the product criterion's shuffle draw is broadcast from rank 0
    torch.manual_seed(rank)
    crit = CGDLoss()
    perm = crit._draw_perm_host(150, 1000)
    digest = torch.cat([p.detach().reshape(-1)[:50] for p in model.student.parameters()])
    q.put((rank, digest.numpy().copy(), logs, perm.tolist(), float(data._pool[0][0].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_kd_engine_two_ranks_gloo():
    world, port = 2, 29743
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, d0, l0, p0, s0), (_, d1, l1, p1, s1) = res
    assert np.array_equal(d0, d1)       # replicas stay bit-identical: same broadcast start, same averaged gradients
    assert l0 == l1                     # log vars are the cross-rank means on every rank
    assert p0 == p1 and sorted(p0) == list(range(150))
    assert s0 != s1                     # ...although each rank trained on its own shard of the data


def test_checkpoint_resume_restores_iteration_and_distillation_step(tmp_path):
    """Q4 of SURVEY section 3.4: the reference loses the KD step counter on resume; here iter, cnt, the student,
    the optimizer state survive a save / resume round trip and the teacher is not part of the checkpoint."""
    import segdistill_amd
    from oracle.eager_modules import swap_in_eager_criteria
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.engine import KDTrainer, SyntheticADE
    from segdistill_amd.segmentors import sd_module
    segdistill_amd.register_all()
    opt = dict(type='AdamW', lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)

    def make(seed):
        sd_module.SYNTHETIC_WEIGHTS_OK = True
        try:
            torch.manual_seed(seed)
            m = build_segmentor(_tiny_sd_cfg())
        finally:
            sd_module.SYNTHETIC_WEIGHTS_OK = False
        swap_in_eager_criteria(m)
        return m, KDTrainer(m, opt, dict(policy='poly', power=1.0, min_lr=0.0, by_epoch=False))

    m1, t1 = make(0)
    data = SyntheticADE(1, size=(64, 64), device='cpu', pool=1)
    for _ in range(2):
        t1.step(data.next())
    path = str(tmp_path / 'latest.pth')
    t1.save(path)
    ck = torch.load(path, weights_only=False)
    # mmcv's checkpoint layout (what the reference's tooling reads): 'meta' / 'state_dict' / 'optimizer'; state_dict = the student's own keys
    assert set(ck) == {'meta', 'state_dict', 'optimizer', 'distillation_loss'} and ck['meta'] == {'iter': 2, 'cnt': 2}
    assert not any(k.startswith(('teacher', 'student')) for k in ck['state_dict']) and 'backbone.patch_embed1.proj.weight' in ck['state_dict']
    # ... so the trained student loads into a bare segmentor through this repo's own loader (round 1: zero keys matched, silently)
    from segdistill_amd.checkpoint import load_checkpoint
    bare = build_segmentor(copy.deepcopy(_tiny_sd_cfg()['cfg_s']))
    res = load_checkpoint(bare, path, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for a, b in zip(m1.student.parameters(), bare.parameters()):
        assert torch.equal(a, b)
    with pytest.raises(KeyError, match='none of its'):
        load_checkpoint(torch.nn.Linear(3, 3), path)              # a checkpoint that matches nothing is an error, not a silent no-op
    m2, t2 = make(1)                      # different init: everything must come from the checkpoint
    t2.resume(path)
    assert t2.iter == 2 and m2.cnt == 2
    for a, b in zip(m1.student.parameters(), m2.student.parameters()):
        assert torch.equal(a, b)
    s1, s2 = t1.optimizer.state_dict()['state'], t2.optimizer.state_dict()['state']
    assert s1.keys() == s2.keys() and all(torch.equal(s1[k]['exp_avg'], s2[k]['exp_avg']) for k in s1)


def test_extractor_and_loss_dispatch_errors_and_naming():
    import segdistill_amd
    from segdistill_amd.distillation import DistillationLoss, Extractor
    segdistill_amd.register_all()
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.ReLU())
    with pytest.raises(KeyError, match='no module named'):
        Extractor(net, net, [dict(student_layer='nope', teacher_layer='0', loss_name='CDLoss', loss_config={})])
    ex = Extractor(net, copy_net := torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.ReLU()),
                   [dict(student_layer='0', teacher_layer='1', loss_name='CDLoss', loss_config={})])
    x = torch.randn(1, 3, 5, 5)
    ex.train()
    net(x); copy_net(x)
    assert list(ex.student_features) == ['0'] and list(ex.teacher_features) == ['1']
    ex.clear(); ex.eval()
    net(x); copy_net(x)
    assert not ex.student_features and not ex.teacher_features      # taps are recorded only while training (opts.py:67)
    with pytest.raises(KeyError, match='not a registered distillation loss'):
        DistillationLoss([dict(student_layer='a', teacher_layer='b', loss_name='os.system', loss_config={})])  # no eval() of config strings
    dl = DistillationLoss([dict(student_layer='a', teacher_layer='b', loss_name='CGDLoss', loss_config=({'group_size': 4},))])   # 1-tuple config
    assert dl.criteria[0].transform_config['group_size'] == 4
    with pytest.raises(ValueError, match='square map'):
        from segdistill_amd.distillation.opts import _to_nchw
        _to_nchw(torch.zeros(1, 10, 4))
    from segdistill_amd.distillation.opts import _to_nchw
    assert _to_nchw(torch.zeros(2, 16, 7)).shape == (2, 7, 4, 4)


def test_swin_public_checkpoint_adaptations(tmp_path):
    """Loading a Swin checkpoint trained with another window size / stored in the public layouts: the relative-position-bias
    tables are resized bicubically, a [1, L, C] absolute position embedding is re-laid-out, MoBY's 'encoder.' prefix is
    stripped (reference mmcv_custom/checkpoint.py:314-343)."""
    import torch.nn.functional as F
    from segdistill_amd.backbones.swin import SwinTransformer
    from segdistill_amd.checkpoint import load_checkpoint
    torch.manual_seed(0)
    kw = dict(pretrain_img_size=64, embed_dim=16, depths=[1, 1], num_heads=[2, 4], out_indices=(0, 1), ape=True, drop_path_rate=0.)
    src = SwinTransformer(window_size=4, **kw)
    dst = SwinTransformer(window_size=7, **kw)
    sd = {('encoder.' + k): v.clone() for k, v in src.state_dict().items()}
    ape = sd['encoder.absolute_pos_embed']                                   # [1, C, H, W] -> the public [1, L, C] layout
    sd['encoder.absolute_pos_embed'] = ape.permute(0, 2, 3, 1).reshape(1, -1, ape.shape[1])
    sd['projector.weight'] = torch.zeros(3)                                  # MoBY's other branch: dropped with the prefix filter
    path = tmp_path / 'swin_w4.pth'
    torch.save({'model': sd}, path)
    result = load_checkpoint(dst, str(path), strict=False)
    assert not result.unexpected_keys
    assert all(k.endswith('relative_position_index') for k in result.missing_keys)      # window-size buffers: the model keeps its own
    key = 'layers.0.blocks.0.attn.relative_position_bias_table'
    t4, t7 = src.state_dict()[key], dst.state_dict()[key]
    assert t4.shape == (49, 2) and t7.shape == (169, 2)
    want = F.interpolate(t4.permute(1, 0).view(1, 2, 7, 7), size=(13, 13), mode='bicubic').view(2, 169).permute(1, 0)
    assert torch.allclose(t7, want)
    assert torch.equal(dst.state_dict()['absolute_pos_embed'], src.state_dict()['absolute_pos_embed'])
    assert torch.equal(dst.state_dict()['patch_embed.proj.weight'], src.state_dict()['patch_embed.proj.weight'])
    x = torch.randn(1, 3, 64, 64)
    assert all(torch.isfinite(o).all() for o in dst(x))


def test_frozen_derived_cache():
    """Derived weight layouts are cached for frozen parameters only, and invalidated by an in-place update (checkpoint load)."""
    from segdistill_amd.layers import frozen_derived
    calls = []
    w = torch.nn.Parameter(torch.arange(6.).reshape(2, 3), requires_grad=False)

    def relayout():
        calls.append(1)
        return w.detach().t().contiguous()
    a = frozen_derived(w, 't', relayout)
    b = frozen_derived(w, 't', relayout)
    assert a is b and len(calls) == 1 and torch.equal(a, w.t())
    with torch.no_grad():
        w.mul_(2)                                        # what load_state_dict's copy_ does: bumps the version counter
    c = frozen_derived(w, 't', relayout)
    assert c is not a and len(calls) == 2 and torch.equal(c, w.t())
    w.requires_grad_(True)                               # trainable: changes every step -> never cached
    frozen_derived(w, 't', relayout)
    frozen_derived(w, 't', relayout)
    assert len(calls) == 4
    other = torch.nn.Parameter(torch.ones(2), requires_grad=False)
    w.requires_grad_(False)
    d = frozen_derived(w, 'u', lambda: (w.detach()[:, 0] * other).clone(), other)
    with torch.no_grad():
        other.add_(1)
    e = frozen_derived(w, 'u', lambda: (w.detach()[:, 0] * other).clone(), other)
    assert not torch.equal(d, e)                         # a dependency changed


def test_deferred_scope_and_column_sum_host_behaviour():
    """deferred.py without a GPU: scopes nest and always close, nothing is deferred outside a scope, CPU tensors take the ATen sum."""
    from segdistill_amd import deferred
    assert not deferred.enabled()
    x = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    assert torch.equal(deferred.column_sum(x), x.sum(0))
    with deferred.scope():
        assert deferred.enabled()
        with deferred.scope():                      # a nested scope joins the outer one
            assert deferred.enabled()
        assert deferred.enabled()
        assert torch.equal(deferred.column_sum(x), x.sum(0))      # CPU input: summed at once even inside a scope
    assert not deferred.enabled()
    with pytest.raises(ValueError):
        with deferred.scope():
            raise ValueError('boom')
    assert not deferred.enabled()                   # an exception inside the scope still closes it


def test_chained_sync_batchnorm_is_a_plain_syncbn_outside_a_recording():
    """layers.ChainedSyncBatchNorm: torch.nn.SyncBatchNorm's parameters / buffers / state-dict keys; SegmentRecorder.cut just runs the
    collective when nothing is being captured; build_norm_layer gives BatchNorm2d without a multi-rank GPU group."""
    from segdistill_amd.engine import segments
    from segdistill_amd.layers import ChainedSyncBatchNorm, build_norm_layer
    bn, ref = ChainedSyncBatchNorm(8), torch.nn.BatchNorm2d(8)
    assert isinstance(bn, torch.nn.SyncBatchNorm) and bn._segments is None
    assert list(bn.state_dict()) == list(ref.state_dict())
    bn.eval()
    x = torch.randn(2, 8, 3, 3)
    assert torch.allclose(bn(x), ref.eval()(x))     # eval mode / CPU: the parent's forward
    name, layer = build_norm_layer(dict(type='SyncBN', requires_grad=True), 8)
    assert name == 'bn' and type(layer) is torch.nn.BatchNorm2d
    rec = segments.SegmentRecorder()
    ran = []
    rec.cut(lambda: ran.append(1))
    assert ran == [1] and rec.items == [] and rec.cuts == 0
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 1), bn, ChainedSyncBatchNorm(8))
    assert segments.attach(net, rec) == 2 and bn._segments is rec
    assert segments.attach(net, None) == 2 and bn._segments is None


def test_bench_launcher_builds_one_rank_per_gpu(monkeypatch):
    """bench.py --gpus N outside torchrun: the parent only builds and runs the launcher command (it must not initialise the GPU)."""
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, env=None, cwd=None, **kw):
        seen['cmd'], seen['env'] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, 'run', fake_run)
    rc = bench.spawn_ranks(4, ['--gpus', '4', '--steps', '3'])
    assert rc == 7                                             # the children's status is the parent's status
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-4:] == ['--gpus', '4', '--steps', '3']
    assert cmd[-5].endswith('bench.py') and seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    import torch
    assert not torch.cuda.is_initialized()


def test_list_typed_distillation_entries_dispatch_like_the_reference():
    """reference opts.py:91-98: an entry whose layers are LISTS hands (attn_s, v_s, attn_t, v_t, student, teacher, gt, step) to its criterion and
    names the loss `loss_{student_layer[0]}<->{teacher_layer}_{loss_name}`.  No live reference loss has that signature (they are in the commented
    generation of losses.py); a user-registered one must work and must coexist with ordinary entries."""
    import torch.nn as nn
    import segdistill_amd
    from segdistill_amd.builder import DISTILL_LOSSES
    from segdistill_amd.distillation.opts import DistillationLoss, Extractor
    segdistill_amd.register_all()
    seen = {}

    if 'PairProbeLoss' not in DISTILL_LOSSES.module_dict:
        @DISTILL_LOSSES.register_module()
        class PairProbeLoss(nn.Module):
            def __init__(self, weight=1.0):
                super().__init__()
                self.weight = weight

            def forward(self, attn_s, v_s, attn_t, v_t, student, teacher, gt, step):
                seen['args'] = (attn_s, v_s, attn_t, v_t, student, teacher, gt, step)
                return self.weight * ((attn_s - attn_t).pow(2).mean() + (v_s - v_t).pow(2).mean())

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.v = nn.Linear(4, 4), nn.Linear(4, 4)

        def forward(self, x):
            return self.a(x) + self.v(x)
    torch.manual_seed(0)
    student, teacher = Net(), Net()
    entries = [dict(student_layer=['a', 'v'], teacher_layer=['a', 'v'], loss_name='PairProbeLoss', loss_config=dict(weight=2.0))]
    ext = Extractor(student, teacher, entries)
    ext.train()
    dl = DistillationLoss(entries)
    x = torch.randn(3, 4)
    student(x)
    teacher(x)
    out = dl(ext.student_features, ext.teacher_features, torch.zeros(3, 1, 2, 2), 7, student, teacher)
    assert list(out) == ["loss_a<->['a', 'v']_PairProbeLoss"]
    want = 2.0 * ((student.a(x) - teacher.a(x)).pow(2).mean() + (student.v(x) - teacher.v(x)).pow(2).mean())
    assert float(out["loss_a<->['a', 'v']_PairProbeLoss"]) == pytest.approx(float(want), rel=1e-6)
    assert seen['args'][4] is student and seen['args'][5] is teacher and seen['args'][7] == 7


def test_fuse_pairs_rule_which_criteria_may_share_a_pass(monkeypatch):
    """DistillationLoss._fuse_pairs (host logic only: the fused kernel call is replaced by a recorder): two KLDLoss 'channel' entries on the SAME
    taps fuse iff one permutation table can order the slots of both -- the partner has no shuffle of its own and a group size of 1, or neither
    shuffles; entries with an align projection, on other taps, or without a fusable resize never do; keys keep the reference's naming + #k."""
    import segdistill_amd
    from segdistill_amd import ops
    from segdistill_amd.distillation import losses as L
    from segdistill_amd.distillation.opts import DistillationLoss
    segdistill_amd.register_all()
    bil = dict(mode='bilinear', align_corners=False)

    def kld(g, shuffle, resize=True, **kw):
        cfg = dict(alpha=1, tau=1, transform_config={'loss_type': 'channel', 'group_size': g}, **kw)
        if resize:
            cfg['resize_config'] = bil
        if shuffle:
            cfg['shuffle_config'] = {'interval': 1000}
        return cfg

    def entry(s, t, cfg, **kw):
        return dict(student_layer=s, teacher_layer=t, loss_name='KLDLoss', loss_config=cfg, **kw)
    calls = []

    def fake_up2(xs, xt, size, ca, cb, perm):
        calls.append((ca[0], cb[0], perm is not None))
        return torch.tensor(1.0), torch.tensor(2.0)
    monkeypatch.setattr(ops, 'cgd_kl_up2', fake_up2)
    monkeypatch.setattr(L.KLDLoss, 'fused_up_size', lambda self, xs, xt, gt: (64, 64) if self.resize_config else None)
    monkeypatch.setattr(L.KLDLoss, '_prepare', lambda self, x, c, n: (float(self.alpha), None, torch.arange(c) if self.shuffle_config else None))
    monkeypatch.setattr(DistillationLoss, 'entry_loss', lambda self, i, xs, xt, gt, step: torch.tensor(10.0 + i))
    feats = {'a': torch.zeros(2, 6, 16, 16), 'b': torch.zeros(2, 6, 16, 16), 'c': torch.zeros(2, 6, 16, 16)}
    gt = torch.zeros(2, 1, 64, 64)

    def run(entries):
        calls.clear()
        out = DistillationLoss(entries)(feats, feats, gt, 5)
        return [float(v) for v in out.values()], list(out)
    # config 3's pair: CGD with a shuffle + channel-wise KL (g = 1, no shuffle) -> fused, the shuffled criterion leads
    vals, keys = run([entry('a', 'b', kld(8, True)), entry('a', 'b', kld(1, False))])
    assert calls == [(8, 1, True)] and vals == [1.0, 2.0] and len(set(keys)) == 2
    # the same pair listed the other way round: still fused, still led by the shuffled one; values go back to their own entries
    vals, _ = run([entry('a', 'b', kld(1, False)), entry('a', 'b', kld(8, True))])
    assert calls == [(8, 1, True)] and vals == [2.0, 1.0]
    # neither shuffles: any two group sizes
    run([entry('a', 'b', kld(3, False)), entry('a', 'b', kld(7, False))])
    assert calls == [(3, 7, False)]
    # the partner shuffles too, or has g > 1 beside a shuffled lead: two different slot orders -> not fused
    vals, _ = run([entry('a', 'b', kld(8, True)), entry('a', 'b', kld(1, True))])
    assert calls == [] and vals == [10.0, 11.0]
    run([entry('a', 'b', kld(8, True)), entry('a', 'b', kld(4, False))])
    assert calls == []
    # other taps, no resize, or an align projection: entry by entry
    run([entry('a', 'b', kld(8, False)), entry('a', 'c', kld(1, False))])
    assert calls == []
    run([entry('a', 'b', kld(8, False, resize=False)), entry('a', 'b', kld(1, False, resize=False))])
    assert calls == []
    run([entry('a', 'b', kld(8, False), channel_nums=(6, 6)), entry('a', 'b', kld(1, False))])
    assert calls == []
    # three entries on the same taps: one pair fuses, the third runs alone; A/B switch off: nothing fuses
    vals, _ = run([entry('a', 'b', kld(8, True)), entry('a', 'b', kld(1, False)), entry('a', 'b', kld(2, False))])
    assert calls == [(8, 1, True)] and vals == [1.0, 2.0, 12.0]
    monkeypatch.setenv('SEGDISTILL_FUSE_PAIRS', '0')
    vals, _ = run([entry('a', 'b', kld(8, True)), entry('a', 'b', kld(1, False))])
    assert calls == [] and vals == [10.0, 11.0]


def test_cpu_plumbing_mode_is_explicit_and_matches_the_oracle():
    """BASELINE configs[0] (CPU train_step via tools/train.py): with the explicit switch (tools/train.py --cpu-plumbing) the KLDLoss criteria
    evaluate CPU taps with ATen ops -- value equal to the oracle's restatement of losses.py:101-112 on the very taps -- and without it they
    raise (no silent fallback: test_sdmodule_contract_and_engine_step_cpu)."""
    import segdistill_amd
    from oracle import kd_ref
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.distillation import losses as kd_losses
    from segdistill_amd.engine import SyntheticADE
    from segdistill_amd.segmentors import sd_module
    segdistill_amd.register_all()
    sd_module.SYNTHETIC_WEIGHTS_OK = True
    try:
        torch.manual_seed(0)
        model = build_segmentor(_tiny_sd_cfg())
    finally:
        sd_module.SYNTHETIC_WEIGHTS_OK = False
    model.train()
    data = SyntheticADE(1, size=(64, 64), device='cpu', pool=1)
    batch = data.next()
    assert kd_losses.CPU_PLUMBING is False
    with pytest.raises(RuntimeError, match='GPU only'):
        model.train_step(batch, None)
    taps = {}
    inner = model.distillation_loss.forward

    def spy(sf, tf, gt, step, *rest):
        taps['s'], taps['t'], taps['gt'] = sf['decode_head.linear_pred'].detach(), tf['decode_head.linear_pred'].detach(), gt
        return inner(sf, tf, gt, step, *rest)

    model.distillation_loss.forward = spy
    model.cnt = 0
    kd_losses.CPU_PLUMBING = True
    try:
        out = model.train_step(batch, None)
    finally:
        kd_losses.CPU_PLUMBING = False
    kd = {k: v for k, v in out['log_vars'].items() if k.startswith('loss_decode_head.linear_pred<->')}
    assert len(kd) == 2
    out['loss'].backward()                                            # the ATen criteria are differentiable: the harness can step
    assert model.student.decode_head.linear_pred.weight.grad is not None
    size = tuple(taps['gt'].shape[2:])
    want = [float(kd_ref.eager_kld(taps['s'].double(), taps['t'].double(), alpha=3, tau=4, out_size=size, loss_type='channel', group_size=8)),
            float(kd_ref.eager_kld(taps['s'].double(), taps['t'].double(), alpha=1, tau=1, out_size=size, loss_type='channel', group_size=1))]
    for got, ref in zip(kd.values(), want):
        assert got == pytest.approx(ref, rel=1e-5)

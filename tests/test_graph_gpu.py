"""hipGraph replay of the KD step == eager execution, including an alpha schedule and shuffle iterations
(host-side state reaches the captured kernels as data)."""
import copy
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model():
    import segdistill_amd
    from segdistill_amd.builder import build_segmentor
    from segdistill_amd.segmentors import sd_module
    segdistill_amd.register_all()
    norm = dict(type='SyncBN', requires_grad=True)

    def seg(v, ch, e):
        return dict(type='EncoderDecoder', pretrained=None, backbone=dict(type=f'mit_{v}', style='pytorch'),
                    decode_head=dict(type='SegFormerHead', in_channels=ch, in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32], channels=128,
                                     dropout_ratio=1e-12, num_classes=150, norm_cfg=norm, align_corners=False, decoder_params=dict(embed_dim=e),
                                     loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))
    bil = dict(mode='bilinear', align_corners=False)
    cfg = dict(type='SDModule', cfg_s=seg('b0', [32, 64, 160, 256], 256), cfg_t=seg('b0', [32, 64, 160, 256], 256),
               distillation=[dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='KLDLoss',
                                  loss_config=dict(alpha=3, tau=4, resize_config=bil, shuffle_config={'interval': 3},
                                                   transform_config={'loss_type': 'channel', 'group_size': 8},
                                                   warmup_config={'mode': 'linear', 'warmup_iters': 5}))],
               t_pretrain=None, train_cfg=dict(), test_cfg=dict(mode='whole'))
    sd_module.SYNTHETIC_WEIGHTS_OK = True
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            torch.manual_seed(0)
            m = build_segmentor(cfg)
    finally:
        sd_module.SYNTHETIC_WEIGHTS_OK = False
    m.student.backbone.reset_drop_path(0.)
    return m.cuda()


def test_deterministic_switch_makes_graph_replay_bit_identical():
    """tools/train.py --deterministic (reference tools/dist_train.sh:8; engine.set_deterministic): two trainers from the same seed, the same
    batches, whole-step hipGraph replay -- after 5 optimizer steps every student tensor is bit-identical (without the switch MIOpen's
    filter-gradient kernels of the patch-embed convolutions accumulate with float atomics and 157 of 191 tensors differ).  In a process of its
    own, as the switch is used (tools/determinism_probe.py: the library searches of a process' first steps run in a throw-away trainer first)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'determinism_probe.py'), '--deterministic', '--size', '256', '--steps', '5'],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'deterministic=True steps=5: 0 of' in r.stdout, r.stdout[-1000:]


@pytest.mark.parametrize('cfg', ['cfg2_segformer_b2_b0_cgd', 'cfg5_segformer_b4_b1_multistage_bf16'])
def test_default_step_is_bit_identical_without_the_switch(cfg):
    """Round 6: no MIOpen convolution is left in the SegFormer configs' step (the patch embeddings are window gather + token GEMM,
    csrc/patch_embed.hip), so two runs from one seed agree bit for bit WITHOUT --deterministic -- the reference needs the switch
    (tools/dist_train.sh:8) and pays for it with MIOpen's naive kernels."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'determinism_probe.py'), '--config', os.path.join(root, 'configs', 'kd', cfg + '.py'),
                        '--size', '256', '--steps', '5'], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'deterministic=False steps=5: 0 of' in r.stdout, r.stdout[-1000:]


@pytest.mark.parametrize('mode', ['full', 'hybrid', 'eager_prefetch'])
def test_graph_replay_matches_eager(mode):
    from segdistill_amd.engine import KDTrainer, SyntheticADE
    opt = dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    ref = _model()
    gra = copy.deepcopy(ref)
    t_e = KDTrainer(ref, opt, dict(policy='poly', power=1.0, min_lr=0.0, by_epoch=False))
    t_g = KDTrainer(gra, opt, dict(policy='poly', power=1.0, min_lr=0.0, by_epoch=False))
    data_e = SyntheticADE(2, size=(128, 128), device='cuda:0', pool=3, seed=1)
    data_g = SyntheticADE(2, size=(128, 128), device='cuda:0', pool=3, seed=1)
    example = dict(img=data_g._pool[0][0], img_metas=None, gt_semantic_seg=data_g._pool[0][1])
    if mode != 'eager_prefetch':
        assert (t_g.enable_graph(example) if mode == 'full' else t_g.enable_hybrid_graph(example)), getattr(t_g, 'graph_error', None)
    assert gra.cnt == 0
    perms = []
    cur_g = data_g.next()
    for it in range(7):                         # crosses the warm-up end (5) and two shuffle iterations (3, 6)
        torch.manual_seed(100 + it)            # the shuffle draws torch.randperm from the CPU RNG
        oe = t_e.step(data_e.next())
        pe = ref.distillation_loss.criteria[0].last_perm
        # two trainers in one process is a test-only situation: the eager one's side-stream work (memsets among it) must have drained before
        # the other's graph -- which holds memset NODES -- replays (the runtime hazard of engine/trainer.py::_issues_memsets)
        torch.cuda.synchronize()
        torch.manual_seed(100 + it)
        nxt_g = data_g.next()
        og = t_g.step(cur_g, nxt_g if it != 4 else None)   # teacher one batch ahead (and one iteration without a hint)
        cur_g = nxt_g
        ve, vg = t_e.log_values(), t_g.log_values()
        assert list(ve) == list(vg)
        for k in ve:
            # 7 AdamW steps of fp32 drift.  Accuracy: the two trainers' weights differ in the last bits after the first step (MIOpen's filter-gradient
            # kernels of the patch-embed convolutions accumulate with atomics), and ONE argmax that flips at the logits' resolution is a 4 x 4 block
            # = 16 output pixels -- four such cells are allowed (round 4: the 8-pixel bound of earlier rounds failed once in ~8 full-suite runs)
            tol = 100.0 * 64 / (2 * 128 * 128) if 'acc' in k else 3e-4 * max(1.0, abs(ve[k]))
            assert vg[k] == pytest.approx(ve[k], abs=tol), (it, k, ve[k], vg[k])
        assert ref.cnt == gra.cnt == it + 1
        assert ref.distillation_loss.criteria[0].alpha == pytest.approx(gra.distillation_loss.criteria[0].alpha)
        perms.append(pe)
    assert perms[2] is not None                 # iteration 3 drew a permutation
    # after 7 optimizer steps the two students still agree closely.  Left out: parameters whose gradient is mathematically ZERO -- a bias in front
    # of a normalisation (the head's linear_c*.proj.bias before linear_fuse's BatchNorm, and backbone.norm4.bias feeding only linear_c4): what
    # arrives there is rounding noise (|g| ~ 1e-9 next to 1e-3 for the weights), AdamW normalises it to full-size steps of noise-determined sign,
    # and the hybrid mode's graphed backbone sums its weight gradients in another order than the eager step's grouped launches (round 5)
    # The exclusion is BY NAME (a magnitude threshold would silently drop a real divergence in a small-gradient parameter), and the named
    # gradients are checked to be noise next to their layer's weight gradient.
    zero_grad = {f'decode_head.linear_c{i}.proj.bias' for i in (1, 2, 3, 4)} | {'backbone.norm4.bias'}
    named_r, named_g = dict(ref.student.named_parameters()), dict(gra.student.named_parameters())
    assert zero_grad <= set(named_r)
    for n in zero_grad:
        w = named_r[n.replace('.bias', '.weight')]
        if named_r[n].grad is not None and w.grad is not None:
            assert float(named_r[n].grad.abs().max()) <= 1e-4 * float(w.grad.abs().max()) + 1e-7, n
    pairs = [(named_r[n], named_g[n]) for n in named_r if n not in zero_grad]
    num = sum(float((a - b).pow(2).sum()) for a, b in pairs)
    den = sum(float(a.pow(2).sum()) for a, _ in pairs)
    assert (num / den) ** 0.5 < 3e-4


@pytest.mark.parametrize('mode', ['full', 'hybrid', 'eager_prefetch'])
def test_bf16_graph_modes_match_eager(mode):
    """precision=dict(activations='bf16') (BASELINE config 5): the teacher graph / prefetch must run under the same
    autocast as the student, whichever mode launches it; the first iteration is bit-comparable up to bf16 reordering."""
    from segdistill_amd.engine import KDTrainer, SyntheticADE
    opt = dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    prec = dict(activations='bf16')
    ref = _model()
    gra = copy.deepcopy(ref)
    t_e = KDTrainer(ref, opt, None, precision=prec)
    t_g = KDTrainer(gra, opt, None, precision=prec)
    assert t_g.bf16 and gra.activation_dtype == torch.bfloat16
    data_e = SyntheticADE(2, size=(128, 128), device='cuda:0', pool=3, seed=1)
    data_g = SyntheticADE(2, size=(128, 128), device='cuda:0', pool=3, seed=1)
    example = dict(img=data_g._pool[0][0], img_metas=None, gt_semantic_seg=data_g._pool[0][1])
    if mode != 'eager_prefetch':
        assert (t_g.enable_graph(example) if mode == 'full' else t_g.enable_hybrid_graph(example))
    cur_g = data_g.next()
    for it in range(3):
        torch.manual_seed(100 + it)
        t_e.step(data_e.next())
        torch.cuda.synchronize()               # as above: drain the eager trainer before the other's graph replays
        torch.manual_seed(100 + it)
        nxt_g = data_g.next()
        t_g.step(cur_g, nxt_g)
        cur_g = nxt_g
        ve, vg = t_e.log_values(), t_g.log_values()
        assert list(ve) == list(vg)
        for k in ve:
            tol = 100.0 * 64 / (2 * 128 * 128) if 'acc' in k else 2e-2 * max(1.0, abs(ve[k]))
            assert vg[k] == pytest.approx(ve[k], abs=tol), (it, k, ve[k], vg[k])


@pytest.mark.parametrize('mode', ['full', 'hybrid', 'eager'])
def test_a_torch_optimizer_keeps_derived_weights_fresh(mode, monkeypatch):
    """Only HipAdamW rewrites the pre-split weight planes itself.  ADVICE r3 (high): under graph replay with a torch optimizer
    (SEGDISTILL_HIP_ADAMW=0, amsgrad, SGD) the captured GEMMs kept multiplying by the capture-time planes.  Round 4 found the eager half: torch's
    FUSED optimizers do not bump the parameters' version counters, so planes.get() never noticed the update either.  build_optimizer now gives
    every torch optimizer a step post-hook that rewrites planes (and bf16 shadows) in place.  Checked directly -- planes == a fresh split of the
    live weight, bit for bit -- and through the losses against a HipAdamW eager run (the two optimizers agree to ~1e-6 per step)."""
    from segdistill_amd import planes
    from segdistill_amd.decode_heads import segformer_head
    from segdistill_amd.engine import KDTrainer, SyntheticADE
    from segdistill_amd.engine.optim import HipAdamW
    # the head's four fuse blocks are this small model's planes entries: keep them as products of their own (round 6 folds them into the branch
    # Linears while training, and a folded weight is a per-step temporary without planes)
    monkeypatch.setattr(segformer_head, '_FOLD_TRAIN', False)
    opt = dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    ref = _model()
    gra = copy.deepcopy(ref)
    t_e = KDTrainer(ref, opt, None)
    monkeypatch.setenv('SEGDISTILL_HIP_ADAMW', '0')
    t_g = KDTrainer(gra, opt, None)
    assert isinstance(t_e.optimizer, HipAdamW) and not isinstance(t_g.optimizer, HipAdamW)
    size = (256, 256)                           # 8192 stage-1 tokens: enough tiles for the planes GEMMs to be dispatched
    data_e = SyntheticADE(2, size=size, device='cuda:0', pool=3, seed=1)
    data_g = SyntheticADE(2, size=size, device='cuda:0', pool=3, seed=1)
    example = dict(img=data_g._pool[0][0], img_metas=None, gt_semantic_seg=data_g._pool[0][1])
    if mode != 'eager':
        assert (t_g.enable_graph(example) if mode == 'full' else t_g.enable_hybrid_graph(example))
    for it in range(5):
        torch.manual_seed(100 + it)
        t_e.step(data_e.next())
        torch.manual_seed(100 + it)
        t_g.step(data_g.next())
        torch.cuda.synchronize()
        mine = {id(p) for p in gra.parameters()}
        ents = [e for e in planes._ENTRIES.values() if e.base() is not None and id(e.base()) in mine]
        assert len(ents) >= 2, 'the student registered too few pre-split planes: the test does not test anything'
        for e in ents:
            for d in ('fwd', 'bwd', 'rows'):
                buf = getattr(e, d)
                if buf is None:
                    continue
                have = buf.clone()
                planes._launch([e])             # a fresh split of the LIVE weight into the same buffers
                torch.cuda.synchronize()
                assert torch.equal(have, getattr(e, d)), (it, d, e.out, e.inp)
        ve, vg = t_e.log_values(), t_g.log_values()
        for k in ve:
            if 'acc' not in k:
                assert vg[k] == pytest.approx(ve[k], abs=3e-4 * max(1.0, abs(ve[k]))), (it, k, ve[k], vg[k])


def test_step_issues_no_memset_and_replays_survive_eager_memsets():
    """ROCm 7.x: a memset NODE of a replayed hipGraph stops zeroing once eager hipMemsetAsync calls have been issued between replays (round 6: one
    `torch.equal` between two steps made every later `decode.loss_seg` read 0.0115 instead of 4.82 -- aten::mean's semaphores were no longer
    cleared).  The step therefore issues NO memset at all (segmentors/base.py::_mean), and replays keep matching the eager run whatever is memset
    in between."""
    import ctypes
    from segdistill_amd.engine import KDTrainer, SyntheticADE
    from segdistill_amd.engine.trainer import _issues_memsets
    opt = dict(type='AdamW', lr=6e-5, betas=(0.9, 0.999), weight_decay=0.01)
    ref = _model()
    gra = copy.deepcopy(ref)
    t_e, t_g = KDTrainer(ref, opt, None), KDTrainer(gra, opt, None)
    size = (256, 256)
    data_e = SyntheticADE(2, size=size, device='cuda:0', pool=3, seed=1)
    data_g = SyntheticADE(2, size=size, device='cuda:0', pool=3, seed=1)
    probe = copy.deepcopy(ref)
    t_p = KDTrainer(probe, opt, None)
    batch = SyntheticADE(2, size=size, device='cuda:0', pool=1, seed=1).next()
    t_p.step(batch)
    assert not _issues_memsets(lambda: t_p.step(batch)), 'the KD step issues a hipMemset again: it would be a memset node of the captured step'
    del t_p, probe
    example = dict(img=data_g._pool[0][0], img_metas=None, gt_semantic_seg=data_g._pool[0][1])
    assert t_g.enable_graph(example)
    hip = ctypes.CDLL('libamdhip64.so')
    scratch = torch.zeros(1 << 16, dtype=torch.uint8, device='cuda:0')
    for it in range(6):
        torch.manual_seed(100 + it)
        t_e.step(data_e.next())
        torch.manual_seed(100 + it)
        t_g.step(data_g.next())
        torch.cuda.synchronize()
        for _ in range(it):                               # 0, 1, 2, ... eager memsets between replays: every phase of the runtime's staging ring
            hip.hipMemsetAsync(ctypes.c_void_p(scratch.data_ptr()), 0, ctypes.c_size_t(64), None)
        assert bool(torch.equal(scratch[:64], torch.zeros_like(scratch[:64])))     # an eager multi-block reduction on top
        ve, vg = t_e.log_values(), t_g.log_values()
        for k in ve:
            if 'acc' not in k:
                assert vg[k] == pytest.approx(ve[k], abs=3e-4 * max(1.0, abs(ve[k]))), (it, k, ve[k], vg[k])


def test_memset_free_mean_matches_aten():
    from segdistill_amd.segmentors.base import _mean
    torch.manual_seed(3)
    for shape in ((2, 256, 256), (8, 512, 512), (3, 100, 100), (5,)):
        x = torch.randn(*shape, device='cuda:0', requires_grad=True)
        y = x.detach().clone().requires_grad_(True)
        a, b = _mean(x), y.mean()
        assert float((a - b).abs()) <= 2e-7 * max(1.0, float(b.abs())) + 1e-8
        a.backward()
        b.backward()
        assert float((x.grad - y.grad).abs().max()) <= 1e-7 * float(y.grad.abs().max())

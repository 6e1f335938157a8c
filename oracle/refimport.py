"""Import pieces of the *reference* (read-only at /root/reference) in the build
container, to generate golden vectors and to validate oracle/kd_ref.py.

TEST INFRASTRUCTURE.  Never used on the GPU box (the reference does not travel)
and never imported by segdistill_amd.

The reference's ``mmseg`` package cannot be imported as a package
(``mmseg/__init__.py:1`` imports the absent ``mmcv``), so individual source
files are loaded BY PATH after registering bare parent packages in
``sys.modules``.  Third-party packages the reference needs but which are not
installed here (mmcv-full 1.2.2, timm 0.3.2, IPython) are replaced by the
minimal behavioural stand-ins in ``oracle/_thirdparty_stubs.py``: they are
stand-ins for *third-party dependencies*, not for any reference source, and
the KD-loss file itself needs none of them (torch only).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF = os.environ.get('SEGDISTILL_REFERENCE', '/root/reference')


def available() -> bool:
    return os.path.isfile(os.path.join(REF, 'mmseg/models/distillation/losses.py'))


def _bare(name):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = []  # mark as package
        sys.modules[name] = m
    return sys.modules[name]


def _load(modname, relpath):
    if modname in sys.modules and getattr(sys.modules[modname], '__file__', None):
        return sys.modules[modname]
    sys.dont_write_bytecode = True  # keep /root/reference free of __pycache__
    path = os.path.join(REF, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    parent, _, child = modname.rpartition('.')
    if parent:
        setattr(_bare(parent), child, mod)
    spec.loader.exec_module(mod)
    return mod


def _cpu_cuda_shim():
    """losses.py:56 allocates the pad slab with ``.cuda()``; on a
    GPU-less host make Tensor.cuda the identity (semantics unchanged)."""
    import torch
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self


def load_losses():
    """-> module object of the reference's distillation/losses.py."""
    assert available(), f'reference not found under {REF}'
    _cpu_cuda_shim()
    for p in ('mmseg', 'mmseg.ops', 'mmseg.models', 'mmseg.models.distillation'):
        _bare(p)
    wr = _load('mmseg.ops.wrappers', 'mmseg/ops/wrappers.py')
    sys.modules['mmseg.ops'].resize = wr.resize
    sys.modules['mmseg.ops'].Upsample = wr.Upsample
    return _load('mmseg.models.distillation.losses', 'mmseg/models/distillation/losses.py')


def load_full():
    """Load the KD segmentor stack of the reference (SDModule, EncoderDecoder,
    MiT, ResNetV1c, SegFormerHead, PSPHead, FCNHead, Swin, UPerHead).
    Returns a namespace with the registries and builder functions."""
    from . import _thirdparty_stubs
    _thirdparty_stubs.install()
    losses = load_losses()
    for p in ('mmseg.core', 'mmseg.core.utils', 'mmseg.utils', 'mmseg.models.losses', 'mmseg.models.utils',
              'mmseg.models.backbones', 'mmseg.models.decode_heads', 'mmseg.models.segmentors', 'mmcv_custom'):
        _bare(p)
    misc = _load('mmseg.core.utils.misc', 'mmseg/core/utils/misc.py')
    core = sys.modules['mmseg.core']
    core.add_prefix = misc.add_prefix
    core.build_pixel_sampler = lambda cfg, **kw: None
    import logging
    sys.modules['mmseg.utils'].get_root_logger = lambda *a, **k: logging.getLogger('mmseg')
    sys.modules['mmcv_custom'].load_checkpoint = lambda *a, **k: None
    builder = _load('mmseg.models.builder', 'mmseg/models/builder.py')
    models = sys.modules['mmseg.models']
    models.builder = builder
    for k in ('BACKBONES', 'NECKS', 'HEADS', 'LOSSES', 'SEGMENTORS', 'build_backbone', 'build_head', 'build_loss',
              'build_neck', 'build_segmentor'):
        setattr(models, k, getattr(builder, k))
    lu = _load('mmseg.models.losses.utils', 'mmseg/models/losses/utils.py')
    acc = _load('mmseg.models.losses.accuracy', 'mmseg/models/losses/accuracy.py')
    lp = sys.modules['mmseg.models.losses']
    lp.accuracy, lp.Accuracy = acc.accuracy, acc.Accuracy
    lp.weight_reduce_loss, lp.weighted_loss = lu.weight_reduce_loss, lu.weighted_loss
    ce = _load('mmseg.models.losses.cross_entropy_loss', 'mmseg/models/losses/cross_entropy_loss.py')
    lp.CrossEntropyLoss = ce.CrossEntropyLoss
    rl = _load('mmseg.models.utils.res_layer', 'mmseg/models/utils/res_layer.py')
    sys.modules['mmseg.models.utils'].ResLayer = rl.ResLayer
    sys.modules['mmseg.models.utils'].__all__ = ['ResLayer']
    _load('mmseg.models.backbones.mix_transformer', 'mmseg/models/backbones/mix_transformer.py')
    _load('mmseg.models.backbones.resnet', 'mmseg/models/backbones/resnet.py')
    _load('mmseg.models.backbones.swin_transformer', 'mmseg/models/backbones/swin_transformer.py')
    _load('mmseg.models.decode_heads.decode_head', 'mmseg/models/decode_heads/decode_head.py')
    _load('mmseg.models.decode_heads.segformer_head', 'mmseg/models/decode_heads/segformer_head.py')
    _load('mmseg.models.decode_heads.psp_head', 'mmseg/models/decode_heads/psp_head.py')
    _load('mmseg.models.decode_heads.fcn_head', 'mmseg/models/decode_heads/fcn_head.py')
    _load('mmseg.models.decode_heads.uper_head', 'mmseg/models/decode_heads/uper_head.py')
    opts = _load('mmseg.models.distillation.opts', 'mmseg/models/distillation/opts.py')
    _load('mmseg.models.segmentors.base', 'mmseg/models/segmentors/base.py')
    _load('mmseg.models.segmentors.encoder_decoder', 'mmseg/models/segmentors/encoder_decoder.py')
    sd = _load('mmseg.models.segmentors.SD_structure', 'mmseg/models/segmentors/SD_structure.py')
    ns = types.SimpleNamespace(builder=builder, losses=losses, opts=opts, SDModule=sd.SDModule,
                               build_segmentor=builder.build_segmentor)
    return ns

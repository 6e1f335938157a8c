"""segdistill_amd -- MI355X-native knowledge-distillation train step for semantic
segmentation (the teacher->student KD hot path of wzpscott/SegDistill).

The dense distillation criteria run as hand-written HIP kernels for gfx950 behind a C ABI
(include/segdistill_hip.h, segdistill_amd/csrc); the networks run on PyTorch-ROCm.
"""
__version__ = '0.1.0'


def register_all():
    """Import every module that registers a class (backbones, heads, losses, segmentors,
    distillation criteria) so that configs can be built by name."""
    from . import backbones, decode_heads, distillation, losses, segmentors  # noqa: F401

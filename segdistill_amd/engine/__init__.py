from .data import SyntheticADE  # noqa: F401
from .dp import DataParallelReducer, init_distributed  # noqa: F401
from .optim import PolyLR, build_optimizer  # noqa: F401
from .trainer import KDTrainer  # noqa: F401


def set_deterministic(flag=True):
    """The reference launches with `--deterministic` (tools/dist_train.sh:8 -> mmseg set_random_seed: cudnn.deterministic = True, benchmark = False).
    Here: MIOpen is held to its deterministic convolution algorithms (the patch-embed filter gradients otherwise accumulate with float atomics and
    two runs diverge in the last bits after one step), its auto-tuner is off, and torch flags every op without a deterministic implementation.
    The HIP kernels of this package are run-to-run deterministic by construction (fixed-order reductions, no float atomics)."""
    import torch
    torch.backends.cudnn.deterministic = bool(flag)
    torch.backends.cudnn.benchmark = not flag
    torch.use_deterministic_algorithms(bool(flag), warn_only=True)
    # torch then also FILLS every torch.empty() with NaN (torch.utils.deterministic.fill_uninitialized_memory: ~790 fill launches per config-2 step,
    # 5.4 ms); every buffer this package allocates with empty() is fully written by the kernel it is handed to, so the fills are switched off
    try:
        torch.utils.deterministic.fill_uninitialized_memory = False
    except AttributeError:
        pass

from .data import SyntheticADE  # noqa: F401
from .dp import DataParallelReducer, init_distributed  # noqa: F401
from .optim import PolyLR, build_optimizer  # noqa: F401
from .trainer import KDTrainer  # noqa: F401

"""Real-data adjacency of the KD train step: the ADE20K training pipeline of the reference
(`local_configs/_base_/datasets/ade20k_repeat.py:7-18`) as numpy transforms + a torch DataLoader with per-rank sharding.

Image decode uses Pillow; geometry / colour arithmetic is restated in numpy with OpenCV's conventions (the reference goes
through mmcv -> cv2, which is not installed here): half-pixel-centre bilinear without antialiasing, floor-rule nearest,
8-bit HSV with H in [0,180).  cv2's fixed-point rounding is NOT reproduced bit for bit (unpinned: no cv2 in the build
container); everything else -- sampling distributions, label mapping, crop / flip / pad semantics, key names -- is the reference's.
"""
from .datasets import DATASETS, ADE20KDataset, CustomDataset, RepeatDataset, build_dataloader, build_dataset  # noqa: F401
from .pipelines import PIPELINES, Compose  # noqa: F401

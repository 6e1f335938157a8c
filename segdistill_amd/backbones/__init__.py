from .mit import MixVisionTransformer  # noqa: F401  (registers mit_b0..mit_b5)
from .resnet import ResNet, ResNetV1c, ResNetV1d  # noqa: F401
from .swin import SwinTransformer  # noqa: F401

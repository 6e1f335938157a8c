from .base import BaseSegmentor  # noqa: F401
from .encoder_decoder import EncoderDecoder  # noqa: F401
from .sd_module import SDModule  # noqa: F401

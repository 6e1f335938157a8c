from .cross_entropy import CrossEntropyLoss, accuracy, cross_entropy  # noqa: F401

from .losses import ATLoss, CDLoss, CGDLoss, CGDLossWS, IFVDLoss, KLDLoss, PDLoss  # noqa: F401
from .opts import DistillationLoss, Extractor, FeatureAlign  # noqa: F401

# Same export list and ORDER as the reference (losses/__init__.py:1-5): component losses first,
# LossWrapper last, because LossWrapper resolves component names from this package.
from .LovaszSoftmax import LovaszSoftmax
from .DenseContrastiveLossV2 import DenseContrastiveLossV2
from .DenseContrastiveLossV2_ms import DenseContrastiveLossV2_ms
from .TwoScaleLoss import TwoScaleLoss
from .LossWrapper import LossWrapper

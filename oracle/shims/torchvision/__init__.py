"""Stand-in for the parts of torchvision the reference hot path touches
(models/encoder.py:72 `torchvision.models.resnet50(pretrained=True)`).
Random weights: there is no network for the ImageNet checkpoint."""
from . import models  # noqa: F401
from . import transforms  # noqa: F401

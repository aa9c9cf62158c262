from .fcos import Scale, FCOSHead, FCOSModule, FCOSOverNeRF  # noqa: F401
from .inference import FCOSPostProcessor  # noqa: F401
from .loss import FCOSLossComputation, IOULoss, RotatedIOULoss  # noqa: F401

from .anchor import AnchorGenerator3D, RPNHead  # noqa: F401
from .feature_extractor import VGG_FPN, ResNet_FPN_256, Bottleneck, SwinTransformer_FPN  # noqa: F401
from .nerf_rpn import NeRFRegionProposalNetwork  # noqa: F401
from .rpn import RegionProposalNetwork  # noqa: F401

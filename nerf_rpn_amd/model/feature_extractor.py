"""3D backbones on the HIP kernels.  Mirrors the constructor signatures, attribute names and state-dict keys of
reference nerf_rpn/model/feature_extractor.py (VGG_FPN :288-377); input [N,4,W,L,H] fp32, output 4 maps [N,256,.,.,.]
(channels-last-backed views).  ``compute_dtype`` (fp32 for parity, bf16 for throughput) is an extra attribute.

ResNet / Swin families: not built yet (SURVEY.md section 8a rows a5, a6 -- later rounds)."""
from typing import Dict, List, Union, cast

import torch
from torch import nn

from .. import ops
from . import hip_nn
from .fpn import FPN

vgg_cfgs: Dict[str, List[Union[str, int]]] = {
    "AF": [64, 128, "F", 256, 256, "M", "F", 512, 512, "M", "F", 512, 512, "M", "F"],
    "DF": [64, 64, 128, 128, "F", 256, 256, 256, "M", "F", 512, 512, 512, "M", "F", 512, 512, 512, "M", "F"],
    "EF": [64, 64, 128, 128, "F", 256, 256, 256, 256, "M", "F", 512, 512, 512, 512, "M", "F", 512, 512, 512, 512, "M", "F"],
}


class VGG_FPN(nn.Module):
    def __init__(self, cfg: str = "AF", in_channels: int = 4, batch_norm: bool = True, input_size: int = 256,
                 conv_at_start: bool = False):
        super().__init__()
        if conv_at_start:
            raise NotImplementedError("conv_at_start is unused by run_rpn.py and has no HIP path")
        if in_channels != 4:
            raise NotImplementedError("the HIP stem kernel is specialised for 4-channel rgb-sigma grids")
        self.out_channels = 256
        self.compute_dtype = torch.float32
        self.layers = self._make_layers(vgg_cfgs[cfg], in_channels, batch_norm, input_size)
        self.fpn_neck = FPN([128, 256, 512, 512], self.out_channels, 4)
        self.conv_at_start = False
        self.starting_layers = None
        self.ds_layers = None

    @staticmethod
    def _make_layers(cfg, in_channels, batch_norm, input_size):
        big = input_size >= 160   # stride-2 stem + 3/2/1 max-pool for large grids (feature_extractor.py:335-343)
        layers: List[nn.Module] = [nn.Conv3d(in_channels, 64, kernel_size=7, stride=2 if big else 1, padding=3),
                                   nn.BatchNorm3d(64), nn.ReLU(inplace=True)]
        if big:
            layers.append(nn.MaxPool3d(kernel_size=3, stride=2, padding=1))
        block: List[nn.Module] = []
        width = 64
        for v in cfg:
            if v == "M":
                block.append(nn.MaxPool3d(kernel_size=2, stride=2, ceil_mode=True))
            elif v == "F":
                layers.append(nn.Sequential(*block))
                block = []
            else:
                v = cast(int, v)
                block.append(nn.Conv3d(width, v, kernel_size=3, padding=1))
                if batch_norm:
                    block.append(nn.BatchNorm3d(v))
                block.append(nn.ReLU(inplace=True))
                width = v
        return nn.Sequential(*layers)

    def forward_cl(self, x):
        """x: channels-last [N,W,L,H,4] in compute dtype -> 4 channels-last maps."""
        mods = list(self.layers)
        stages = [m for m in mods if isinstance(m, nn.Sequential)]
        head = mods[:len(mods) - len(stages)]
        x = hip_nn.run_modules(head, x)
        taps = []
        for st in stages:
            x = hip_nn.run_modules(st, x)
            taps.append(x)
        return self.fpn_neck.forward_cl(taps[-4:])

    def forward(self, X):
        x = ops.to_channels_last(X, self.compute_dtype)
        return tuple(hip_nn.as_ncdhw(o) for o in self.forward_cl(x))


def _unbuilt(name):
    class _Missing(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            raise NotImplementedError(f"{name}: HIP path not built yet (SURVEY.md section 8a); only VGG_FPN is available")
    _Missing.__name__ = name
    return _Missing


ResNet_FPN_256 = _unbuilt("ResNet_FPN_256")
ResNet_FPN_64 = _unbuilt("ResNet_FPN_64")
ResNetSimplified_64 = _unbuilt("ResNetSimplified_64")
ResNetSimplified_256 = _unbuilt("ResNetSimplified_256")
SwinTransformer_FPN = _unbuilt("SwinTransformer_FPN")
Bottleneck = _unbuilt("Bottleneck")

"""3D backbones on the HIP kernels.  Mirrors the constructor signatures, attribute names and state-dict keys of
reference nerf_rpn/model/feature_extractor.py (VGG_FPN :288-377); input [N,4,W,L,H] fp32, output 4 maps [N,256,.,.,.]
(channels-last-backed views).  ``compute_dtype`` (fp32 for parity, bf16 for throughput) is an extra attribute.

ResNet_FPN_256 + Bottleneck: feature_extractor.py:31-68, 145-235.  Swin family: not built yet (row a6 -- later rounds)."""
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


class Bottleneck(nn.Module):
    """ResNet bottleneck for 3D grids; stride sits on the first 1x1x1 conv (reference feature_extractor.py:31-68)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv3d(inplanes, planes, kernel_size=1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm3d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_cl(self, x):
        out = hip_nn.batch_norm(self.bn1, hip_nn.conv3d(self.conv1, x), True)
        out = hip_nn.batch_norm(self.bn2, hip_nn.conv3d(self.conv2, out), True)
        out = hip_nn.batch_norm(self.bn3, hip_nn.conv3d(self.conv3, out), False)
        res = x
        if self.downsample is not None:
            res = hip_nn.batch_norm(self.downsample[1], hip_nn.conv3d(self.downsample[0], x), False)
        return ops.AddReluFn.apply(out, res, True)

    def forward(self, x):
        return hip_nn.as_ncdhw(self.forward_cl(hip_nn.as_ndhwc(x, x.dtype)))


class ResNet_FPN_256(nn.Module):
    """ResNet-50-3D + top-down pyramid (reference feature_extractor.py:145-235): returns 4 maps of 256 channels."""

    def __init__(self, block, layers, input_dim=4, is_max_pool=False):
        super().__init__()
        if input_dim != 4:
            raise NotImplementedError("the HIP stem kernel is specialised for 4-channel rgb-sigma grids")
        self.in_planes = 64
        self.out_channels = 256
        self.compute_dtype = torch.float32
        self.conv1 = nn.Conv3d(input_dim, self.in_planes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm3d(self.in_planes)
        self.layers = nn.ModuleList()
        self.start_deep = self.in_planes
        self.is_max_pool = is_max_pool
        for i, depth in enumerate(layers):
            self.layers.append(self._make_layer(block, self.start_deep * (2 ** i), depth, stride=1 if i == 0 else 2))
        self.smooths = nn.ModuleList(nn.Conv3d(256, 256, kernel_size=3, stride=1, padding=1) for _ in range(len(layers) - 1))
        self.latlayers = nn.ModuleList(nn.Conv3d(block.expansion * self.start_deep * (2 ** i), self.out_channels, kernel_size=1)
                                       for i in range(len(layers) - 1, -1, -1))
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._pool = nn.MaxPool3d(kernel_size=3, stride=2, padding=1)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.in_planes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv3d(self.in_planes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm3d(planes * block.expansion))
        mods = [block(self.in_planes, planes, stride, downsample)]
        self.in_planes = planes * block.expansion
        mods += [block(self.in_planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward_cl(self, x):
        c = hip_nn.batch_norm(self.bn1, hip_nn.conv3d(self.conv1, x), True)
        if self.is_max_pool:
            c = hip_nn.max_pool(self._pool, c)
        taps = []
        for stage in self.layers:
            for blk in stage:
                c = blk.forward_cl(c)
            taps.append(c)
        p = [hip_nn.conv3d(self.latlayers[0], taps[-1])]
        for i in range(len(self.latlayers) - 1):
            lat = hip_nn.conv3d(self.latlayers[i + 1], taps[-2 - i])
            p.append(ops.UpsampleAddFn.apply(lat, p[i]))
        for i, sm in enumerate(self.smooths):
            p[i + 1] = hip_nn.conv3d(sm, p[i + 1])
        p.reverse()
        return p

    def forward(self, x):
        return [hip_nn.as_ncdhw(o) for o in self.forward_cl(ops.to_channels_last(x, self.compute_dtype))]


def _unbuilt(name):
    class _Missing(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            raise NotImplementedError(f"{name}: HIP path not built yet (SURVEY.md section 8a); VGG_FPN and ResNet_FPN_256 are available")
    _Missing.__name__ = name
    return _Missing


ResNet_FPN_64 = _unbuilt("ResNet_FPN_64")
ResNetSimplified_64 = _unbuilt("ResNetSimplified_64")
ResNetSimplified_256 = _unbuilt("ResNetSimplified_256")
SwinTransformer_FPN = _unbuilt("SwinTransformer_FPN")

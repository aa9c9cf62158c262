"""ORACLE (test infrastructure only): backbone / neck / head as plain torch-CPU modules.

State-dict key names equal the reference's so checkpoints and fixtures interchange
(SURVEY.md App. A.4).  Restates reference nerf_rpn/model/feature_extractor.py:273-377 (VGG),
fpn.py:8-185 (FPN with default arguments), anchor.py:177-213 (RPNHead).
"""
import torch
import torch.nn.functional as F
from torch import nn

VGG_CFG = {
    "AF": [64, 128, "F", 256, 256, "M", "F", 512, 512, "M", "F", 512, 512, "M", "F"],
    "EF": [64, 64, 128, 128, "F", 256, 256, 256, 256, "M", "F", 512, 512, 512, 512, "M", "F",
           512, 512, 512, 512, "M", "F"],
}


class FPN(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lateral_convs = nn.ModuleList(nn.Conv3d(c, out_channels, 1) for c in in_channels)
        self.fpn_convs = nn.ModuleList(nn.Conv3d(out_channels, out_channels, 3, padding=1) for _ in in_channels)

    def forward(self, xs):
        lat = [conv(x) for conv, x in zip(self.lateral_convs, xs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        return tuple(conv(x) for conv, x in zip(self.fpn_convs, lat))


class VGGFPN(nn.Module):
    """VGG_FPN(cfg, 4, True, input_size); feature_extractor.py:288-377."""

    def __init__(self, cfg="EF", in_channels=4, input_size=160):
        super().__init__()
        self.out_channels = 256
        layers = [nn.Conv3d(in_channels, 64, 7, stride=2 if input_size >= 160 else 1, padding=3),
                  nn.BatchNorm3d(64), nn.ReLU(inplace=True)]
        if input_size >= 160:
            layers.append(nn.MaxPool3d(3, stride=2, padding=1))
        cur, c_in = [], 64
        for v in VGG_CFG[cfg]:
            if v == "M":
                cur.append(nn.MaxPool3d(2, stride=2, ceil_mode=True))
            elif v == "F":
                layers.append(nn.Sequential(*cur))
                cur = []
            else:
                cur += [nn.Conv3d(c_in, v, 3, padding=1), nn.BatchNorm3d(v), nn.ReLU(inplace=True)]
                c_in = v
        self.layers = nn.Sequential(*layers)
        self.fpn_neck = FPN([128, 256, 512, 512], 256)

    def forward(self, x):
        feats = []
        for layer in self.layers:
            x = layer(x)
            feats.append(x)
        return self.fpn_neck(feats[-4:])


class RPNHead(nn.Module):
    """anchor.py:177-213."""

    def __init__(self, in_channels, num_anchors, conv_depth=1, rotate=False):
        super().__init__()
        seq = []
        for _ in range(conv_depth):
            seq += [nn.Conv3d(in_channels, in_channels, 3, padding=1), nn.ReLU(inplace=True)]
        self.conv = nn.Sequential(*seq)
        self.cls_logits = nn.Conv3d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv3d(in_channels, num_anchors * (8 if rotate else 6), 1)
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.normal_(m.weight, std=0.01)
                nn.init.constant_(m.bias, 0)

    def forward(self, feats):
        logits, deltas = [], []
        for f in feats:
            t = self.conv(f)
            logits.append(self.cls_logits(t))
            deltas.append(self.bbox_pred(t))
        return logits, deltas


class Bottleneck(nn.Module):
    """feature_extractor.py:31-68 (stride on the first 1x1x1 conv)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv3d(inplanes, planes, 1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        r = x if self.downsample is None else self.downsample(x)
        return F.relu(y + r)


class ResNetFPN(nn.Module):
    """ResNet_FPN_256(Bottleneck, layers, 4, is_max_pool); feature_extractor.py:145-235."""

    def __init__(self, layers=(3, 4, 6, 3), is_max_pool=True):
        super().__init__()
        self.out_channels = 256
        self.is_max_pool = is_max_pool
        self.conv1 = nn.Conv3d(4, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        self.layers = nn.ModuleList()
        inp = 64
        for i, depth in enumerate(layers):
            planes, stride = 64 * 2 ** i, (1 if i == 0 else 2)
            ds = None
            if stride != 1 or inp != planes * 4:
                ds = nn.Sequential(nn.Conv3d(inp, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm3d(planes * 4))
            blocks = [Bottleneck(inp, planes, stride, ds)]
            inp = planes * 4
            blocks += [Bottleneck(inp, planes) for _ in range(1, depth)]
            self.layers.append(nn.Sequential(*blocks))
        self.smooths = nn.ModuleList(nn.Conv3d(256, 256, 3, padding=1) for _ in range(len(layers) - 1))
        self.latlayers = nn.ModuleList(nn.Conv3d(4 * 64 * 2 ** i, 256, 1) for i in range(len(layers) - 1, -1, -1))

    def forward(self, x):
        c = F.relu(self.bn1(self.conv1(x)))
        if self.is_max_pool:
            c = F.max_pool3d(c, 3, 2, 1)
        taps = []
        for stage in self.layers:
            c = stage(c)
            taps.append(c)
        p = [self.latlayers[0](taps[-1])]
        for i in range(len(self.latlayers) - 1):
            lat = self.latlayers[i + 1](taps[-2 - i])
            p.append(F.interpolate(p[i], size=lat.shape[2:], mode="nearest") + lat)
        for i, sm in enumerate(self.smooths):
            p[i + 1] = sm(p[i + 1])
        return p[::-1]

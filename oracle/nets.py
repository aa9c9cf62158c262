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

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


# ----------------------------------------------------------------------------------------------------------------------
# Swin-3D (feature_extractor.py:382-789).  Written with explicit pad / roll / window partition on [B,H,W,D,C] tokens.
# ----------------------------------------------------------------------------------------------------------------------
def window_attention(x, qkv_w, qkv_b, proj_w, proj_b, table, index, heads, shift, ws=4):
    """shifted_window_attention (feature_extractor.py:382-500) for a cubic window ``ws`` and shift in {0, ws // 2}."""
    B, H, W, D, C = x.shape
    pad = [(ws - s % ws) % ws for s in (H, W, D)]
    x = F.pad(x, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    pH, pW, pD = x.shape[1:4]
    sh = [0 if ws >= p else shift for p in (pH, pW, pD)]          # :425-430
    if sum(sh) > 0:
        x = torch.roll(x, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
    nW = (pH // ws) * (pW // ws) * (pD // ws)

    def partition(t):     # [.., pH, pW, pD, c] -> [.. * nW, ws^3, c]
        lead, c = t.shape[:-4], t.shape[-1]
        t = t.reshape(*lead, pH // ws, ws, pW // ws, ws, pD // ws, ws, c)
        n = len(lead)
        t = t.permute(*range(n), n, n + 2, n + 4, n + 1, n + 3, n + 5, n + 6)
        return t.reshape(-1, ws ** 3, c)

    xw = partition(x)
    qkv = F.linear(xw, qkv_w, qkv_b).reshape(xw.shape[0], ws ** 3, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    attn = attn + table[index].view(ws ** 3, ws ** 3, -1).permute(2, 0, 1).unsqueeze(0)
    if sum(sh) > 0:       # :463-479
        region = x.new_zeros((pH, pW, pD))
        count = 0
        for h in ((0, -ws), (-ws, -sh[0]), (-sh[0], None)):
            for w in ((0, -ws), (-ws, -sh[1]), (-sh[1], None)):
                for d in ((0, -ws), (-ws, -sh[2]), (-sh[2], None)):
                    region[h[0]:h[1], w[0]:w[1], d[0]:d[1]] = count
                    count += 1
        r = partition(region[..., None])[..., 0]                 # [nW, ws^3]
        mask = r.unsqueeze(1) - r.unsqueeze(2)
        mask = torch.where(mask != 0, torch.full_like(mask, -100.0), torch.zeros_like(mask))
        attn = (attn.view(B, nW, heads, ws ** 3, ws ** 3) + mask[None, :, None]).view(-1, heads, ws ** 3, ws ** 3)
    attn = F.softmax(attn, dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(xw.shape[0], ws ** 3, C)
    y = F.linear(y, proj_w, proj_b)
    y = y.view(B, pH // ws, pW // ws, pD // ws, ws, ws, ws, C).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, pH, pW, pD, C)
    if sum(sh) > 0:
        y = torch.roll(y, shifts=(sh[0], sh[1], sh[2]), dims=(1, 2, 3))
    return y[:, :H, :W, :D].contiguous()


class WindowAttention(nn.Module):
    """ShiftedWindowAttention (feature_extractor.py:533-613)."""

    def __init__(self, dim, heads, shift, ws=4):
        super().__init__()
        self.heads, self.shift, self.ws = heads, shift, ws
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 3, heads))
        a = torch.arange(ws)
        c = torch.stack(torch.meshgrid(a, a, a, indexing="ij")).flatten(1)
        rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
        self.register_buffer("relative_position_index",
                             (rel[..., 0] * (2 * ws - 1) ** 2 + rel[..., 1] * (2 * ws - 1) + rel[..., 2]).flatten())

    def forward(self, x):
        return window_attention(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias,
                                self.relative_position_bias_table, self.relative_position_index, self.heads, self.shift, self.ws)


class SwinBlock(nn.Module):
    """SwinTransformerBlock (feature_extractor.py:616-653); ``drop`` = StochasticDepth('row') probability.
    ``noise_hook(shape) -> tensor`` lets a test inject the per-sample keep mask instead of drawing it."""

    def __init__(self, dim, heads, shift, drop=0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn = WindowAttention(dim, heads, shift)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.mlp = nn.Sequential(nn.Linear(dim, 4 * dim), nn.GELU(), nn.Dropout(0.0), nn.Linear(4 * dim, dim), nn.Dropout(0.0))
        self.drop = drop
        self.noise_hook = None

    def _sd(self, y):
        if not self.training or self.drop == 0.0:
            return y
        keep = 1.0 - self.drop
        shape = [y.shape[0]] + [1] * (y.ndim - 1)
        noise = self.noise_hook(shape) if self.noise_hook else torch.empty(shape).bernoulli_(keep)
        return y * (noise / keep if keep > 0 else noise)

    def forward(self, x):
        x = x + self._sd(self.attn(self.norm1(x)))
        return x + self._sd(self.mlp(self.norm2(x)))


class PatchMerge(nn.Module):
    """PatchMerging (feature_extractor.py:656-690)."""

    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim, eps=1e-5)

    def forward(self, x):
        H, W, D = x.shape[1:4]
        x = F.pad(x, (0, 0, 0, D % 2, 0, W % 2, 0, H % 2))
        parts = [x[:, i::2, j::2, k::2] for k in (0, 1) for j in (0, 1) for i in (0, 1)]     # x0..x7 of :672-680
        return self.reduction(self.norm(torch.cat(parts, -1)))


class _Permute(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.dims = dims

    def forward(self, x):
        return x.permute(*self.dims)


class SwinFPN(nn.Module):
    """SwinTransformer_FPN(patch 4, window 4, expand_dim) (feature_extractor.py:692-789)."""

    def __init__(self, embed_dim=96, depths=(2, 2, 18, 2), heads=(3, 6, 12, 24), stochastic_depth_prob=0.1):
        super().__init__()
        self.out_channels = 256
        self.patch_partition = nn.Sequential(nn.Conv3d(4, embed_dim, 4, stride=4), _Permute([0, 2, 3, 4, 1]),
                                             nn.LayerNorm(embed_dim, eps=1e-5))
        self.stages = nn.ModuleList()
        total, bid = sum(depths), 0
        for i, depth in enumerate(depths):
            dim = embed_dim * 2 ** i
            mods = [PatchMerge(dim // 2)] if i > 0 else []
            for j in range(depth):
                mods.append(SwinBlock(dim, heads[i], 0 if j % 2 == 0 else 2, stochastic_depth_prob * bid / (total - 1)))
                bid += 1
            self.stages.append(nn.Sequential(*mods))
        self.fpn_neck = FPN([embed_dim * 2 ** i for i in range(len(depths))], 256)

    def forward(self, x):
        x = self.patch_partition(x)
        feats = []
        for stage in self.stages:
            x = stage(x)
            feats.append(x.permute(0, 4, 1, 2, 3).contiguous())
        return self.fpn_neck(feats)

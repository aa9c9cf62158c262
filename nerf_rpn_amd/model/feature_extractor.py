"""3D backbones on the HIP kernels.  Mirrors the constructor signatures, attribute names and state-dict keys of
reference nerf_rpn/model/feature_extractor.py (VGG_FPN :288-377); input [N,4,W,L,H] fp32, output 4 maps [N,256,.,.,.]
(channels-last-backed views).  ``compute_dtype`` (fp32 for parity, bf16 for throughput) is an extra attribute.

ResNet_FPN_256 + Bottleneck: feature_extractor.py:31-68, 145-235.  SwinTransformer_FPN (+ ShiftedWindowAttention,
SwinTransformerBlock, PatchMerging): feature_extractor.py:382-789."""
from functools import partial
from typing import Callable, Dict, List, Optional, Union, cast

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
        counted = self.training
        if counted:
            hip_nn.bn_counters(self).step()      # all 17 num_batches_tracked buffers in one launch
        x = hip_nn.run_modules(head, x, counted)
        taps = []
        for st in stages:
            x = hip_nn.run_modules(st, x, counted)
            taps.append(x)
        return self.fpn_neck.forward_cl(taps[-4:])

    def forward(self, X):
        x = ops.to_channels_last(X, self.compute_dtype)
        return tuple(hip_nn.as_ncdhw(o) for o in self.forward_cl(x))

    def feature_grids(self, size):
        """(X, Y, Z) of the 4 output maps for an input grid ``size`` -- shape arithmetic only, so target assignment can be queued
        before the backbone runs."""
        mods = list(self.layers)
        stages = [m for m in mods if isinstance(m, nn.Sequential)]
        size = hip_nn.module_out_size(mods[:len(mods) - len(stages)], tuple(size))
        grids = []
        for st in stages:
            size = hip_nn.module_out_size(list(st), size)
            grids.append(size)
        return grids[-4:]


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
        out = hip_nn.conv_bn(self.conv1, self.bn1, x, True)
        out = hip_nn.conv_bn(self.conv2, self.bn2, out, True)
        out = hip_nn.conv_bn(self.conv3, self.bn3, out, False)
        res = x
        if self.downsample is not None:
            res = hip_nn.conv_bn(self.downsample[0], self.downsample[1], x, False)
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
        c = hip_nn.conv_bn(self.conv1, self.bn1, x, True)
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

    def feature_grids(self, size):
        size = hip_nn.conv_out(size, 7, 2, 3)
        if self.is_max_pool:
            size = hip_nn.module_out_size([self._pool], size)
        grids = []
        for i in range(len(self.layers)):
            if i > 0:
                size = tuple((g - 1) // 2 + 1 for g in size)        # stride-2 1x1x1 conv of the stage's first bottleneck
            grids.append(size)
        return grids


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


# ======================================================================================================================
# Swin-3D (reference feature_extractor.py:382-789).  Tokens are channels-last [B,H,W,D,C] -- the activation layout of the
# whole HIP path -- so the reference's Permute modules are no-ops here; they stay in the containers to keep state-dict keys.
# ======================================================================================================================
def linear(mod, x, relu=False):
    """nn.Linear on the last dimension of a channels-last token tensor = the 1x1x1 MFMA GEMM."""
    return ops.ConvFn.apply(x, hip_nn._pack_of(mod), mod.out_features, relu, False, 1, mod.weight, mod.bias)


def layer_norm(mod, x):
    return ops.LayerNormFn.apply(x, mod.weight, mod.bias, mod.eps)


class Permute(nn.Module):
    """Parameter-free placeholder of torchvision.ops.misc.Permute (keeps ``patch_partition.2`` as the LayerNorm key)."""

    def __init__(self, dims):
        super().__init__()
        self.dims = dims

    def forward(self, x):
        return torch.permute(x, self.dims)


class StochasticDepth(nn.Module):
    """torchvision.ops.StochasticDepth(p, "row"): per-sample Bernoulli(1-p)/(1-p) scale of a residual branch while training.
    ``scale()`` returns the per-sample factor (None = identity); the multiply is fused into the residual-join kernel."""

    def __init__(self, p: float, mode: str = "row"):
        super().__init__()
        if mode != "row":
            raise NotImplementedError("only StochasticDepth('row') is used by the Swin backbone")
        self.p, self.mode = p, mode

    def scale(self, x):
        if not self.training or self.p == 0.0:
            return None
        keep = 1.0 - self.p
        noise = torch.empty(x.shape[0], dtype=torch.float32, device=x.device).bernoulli_(keep)
        return noise / keep if keep > 0.0 else noise


class ShiftedWindowAttention(nn.Module):
    """Window multi-head self attention with relative position bias (reference :533-613), window 4x4x4, head_dim 32."""

    def __init__(self, dim, window_size, shift_size, num_heads, qkv_bias=True, proj_bias=True, attention_dropout=0.0, dropout=0.0):
        super().__init__()
        if len(window_size) != 3 or len(shift_size) != 3:
            raise ValueError("window_size and shift_size must be of length 3")
        if dim % num_heads != 0:
            # --backbone_type swin_b (run_rpn.py:284: embed_dim 128 with heads 3/6/12/24) cannot run in the reference either: its attention
            # reshapes the 3*C qkv columns to [3, heads, C // heads] (feature_extractor.py:446) and 3 * 3 * 42 != 384 raises in torch
            raise ValueError(f"embed dim {dim} is not divisible by num_heads {num_heads} (the reference's swin_b table, run_rpn.py:284, has this "
                             "defect: its own attention reshape fails)")
        if list(window_size) != [4, 4, 4] or dim != 32 * num_heads or list(shift_size) not in ([0, 0, 0], [2, 2, 2]):
            raise NotImplementedError("the HIP attention kernel is specialised for window 4x4x4, shift 0|2, head_dim 32 "
                                      "(swin_t / swin_s / swin_l of run_rpn.py)")
        if attention_dropout != 0.0 or dropout != 0.0:
            raise NotImplementedError("dropout is 0 in every NeRF-RPN configuration and has no HIP path")
        self.window_size, self.shift_size, self.num_heads = list(window_size), list(shift_size), num_heads
        self.attention_dropout, self.dropout = attention_dropout, dropout
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        w = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * w[0] - 1) * (2 * w[1] - 1) * (2 * w[2] - 1), num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        c = torch.stack(torch.meshgrid(torch.arange(w[0]), torch.arange(w[1]), torch.arange(w[2]), indexing="ij")).flatten(1)
        rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] = (rel[:, :, 0] + w[0] - 1) * (2 * w[2] - 1) * (2 * w[1] - 1)
        rel[:, :, 1] = (rel[:, :, 1] + w[1] - 1) * (2 * w[2] - 1)
        rel[:, :, 2] = rel[:, :, 2] + w[2] - 1
        self.register_buffer("relative_position_index", rel.sum(-1).flatten())     # int64 [4096], checkpoint-compatible

    def _index32(self):
        idx = self.__dict__.get("_nrpn_idx32")
        if idx is None or idx.device != self.relative_position_index.device:
            idx = self.relative_position_index.to(torch.int32).contiguous()
            # the bf16 MFMA attention kernels compute this index arithmetically: make sure a loaded checkpoint agrees
            t = torch.arange(64, device=idx.device)
            code = (t // 16) * 49 + ((t // 4) % 4) * 7 + t % 4
            if not torch.equal(idx.view(64, 64).long(), code[:, None] - code[None, :] + 171):
                raise NotImplementedError("relative_position_index differs from the reference's 4x4x4 window formula")
            self.__dict__["_nrpn_idx32"] = idx
        return idx

    def forward(self, x):
        """x: [B,H,W,D,C] channels-last tokens."""
        qkv = linear(self.qkv, x)
        ctx = ops.WindowAttnFn.apply(qkv, self.qkv.bias, self.relative_position_bias_table, self._index32(), self.num_heads,
                                     sum(self.shift_size) > 0)
        return linear(self.proj, ctx)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, dropout=0.0, attention_dropout=0.0,
                 stochastic_depth_prob=0.0, norm_layer: Callable[..., nn.Module] = nn.LayerNorm,
                 attn_layer: Callable[..., nn.Module] = ShiftedWindowAttention):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = attn_layer(dim, window_size, shift_size, num_heads, attention_dropout=attention_dropout, dropout=dropout)
        self.stochastic_depth = StochasticDepth(stochastic_depth_prob, "row")
        self.norm2 = norm_layer(dim)
        hidden = int(dim * mlp_ratio)
        # torchvision MLP container layout: Linear, GELU, Dropout, Linear, Dropout  (state-dict keys mlp.0.*, mlp.3.*)
        self.mlp = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden, dim), nn.Dropout(dropout))
        for m in self.mlp.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.normal_(m.bias, std=1e-6)

    def forward(self, x):
        a = self.attn(layer_norm(self.norm1, x))
        x = ops.ScaleAddFn.apply(x, a, self.stochastic_depth.scale(x))
        h = ops.GeluFn.apply(linear(self.mlp[0], layer_norm(self.norm2, x)))
        m = linear(self.mlp[3], h)
        return ops.ScaleAddFn.apply(x, m, self.stochastic_depth.scale(x))


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer: Callable[..., nn.Module] = nn.LayerNorm, expand_dim: bool = True):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(8 * dim, dim * 2 if expand_dim else dim, bias=False)
        self.norm = norm_layer(8 * dim)

    def forward(self, x):
        return linear(self.reduction, layer_norm(self.norm, ops.PatchMergeFn.apply(x)))


class SwinTransformer_FPN(nn.Module):
    """3D Swin Transformer + FPN (reference :692-789): patch embedding (k4 s4 conv as gather + GEMM), LayerNorm, 4 stages of
    shifted-window blocks with patch merging between them, FPN over the 4 stage outputs."""

    def __init__(self, patch_size: List[int], embed_dim: int, depths: List[int], num_heads: List[int], window_size: List[int],
                 mlp_ratio: float = 4.0, dropout: float = 0.0, attention_dropout: float = 0.0, stochastic_depth_prob: float = 0.1,
                 norm_layer: Optional[Callable[..., nn.Module]] = partial(nn.LayerNorm, eps=1e-5),
                 block: Optional[Callable[..., nn.Module]] = SwinTransformerBlock,
                 downsample_layer: Callable[..., nn.Module] = PatchMerging, expand_dim: bool = True, out_channels: int = 256,
                 input_dim: int = 4):
        super().__init__()
        if input_dim != 4 or len(set(patch_size)) != 1:
            raise NotImplementedError("the HIP patch embedding is specialised for 4-channel grids and cubic patches")
        self.out_channels = out_channels
        self.compute_dtype = torch.float32
        self.patch_size = patch_size[0]
        self.patch_partition = nn.Sequential(
            nn.Conv3d(input_dim, embed_dim, kernel_size=tuple(patch_size), stride=tuple(patch_size)),
            Permute([0, 2, 3, 4, 1]),
            norm_layer(embed_dim),
        )
        self.stages = nn.ModuleList()
        total_stage_blocks = sum(depths)
        stage_block_id = 0
        fpn_in_channels = []
        for i_stage in range(len(depths)):
            stage: List[nn.Module] = []
            dim = embed_dim * 2 ** i_stage if expand_dim else embed_dim
            fpn_in_channels.append(dim)
            if i_stage > 0:
                stage.append(downsample_layer(fpn_in_channels[-2], norm_layer, expand_dim))
            for i_layer in range(depths[i_stage]):
                sd_prob = stochastic_depth_prob * float(stage_block_id) / (total_stage_blocks - 1)
                stage.append(block(dim, num_heads[i_stage], window_size=window_size,
                                   shift_size=[0 if i_layer % 2 == 0 else w // 2 for w in window_size], mlp_ratio=mlp_ratio,
                                   dropout=dropout, attention_dropout=attention_dropout, stochastic_depth_prob=sd_prob,
                                   norm_layer=norm_layer))
                stage_block_id += 1
            self.stages.append(nn.Sequential(*stage))
        self.fpn_neck = FPN(fpn_in_channels, out_channels, len(fpn_in_channels))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_cl(self, x):
        """x: channels-last [N,W,L,H,4] in compute dtype -> 4 channels-last maps."""
        embed = self.patch_partition[0]
        x = ops.ConvFn.apply(ops.patchify(x, self.patch_size), hip_nn._pack_of(embed), embed.out_channels, False, False, 1,
                             embed.weight, embed.bias)
        x = layer_norm(self.patch_partition[2], x)
        feats = []
        for stage in self.stages:
            for mod in stage:
                x = mod(x)
            feats.append(x)
        return self.fpn_neck.forward_cl(feats)

    def forward(self, X):
        x = ops.to_channels_last(X, self.compute_dtype)
        return tuple(hip_nn.as_ncdhw(o) for o in self.forward_cl(x))

    def feature_grids(self, size):
        size = tuple(int(g) // self.patch_size for g in size)
        grids = [size]
        for _ in range(len(self.stages) - 1):
            size = tuple((g + 1) // 2 for g in size)                # PatchMerging pads odd sizes
            grids.append(size)
        return grids

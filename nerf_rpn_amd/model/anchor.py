"""Anchor generator and RPN head on the HIP kernels.  API mirrors reference nerf_rpn/model/anchor.py:14-213.

Anchors are never materialised on the hot path: an ``ops.AnchorTable`` describes the pyramid and kernels compute each
anchor from its flat (level, x, y, z, a) index.  ``AnchorGenerator3D.forward`` still returns the explicit tensors for
API compatibility.  The head's ``cls_logits`` and ``bbox_pred`` 1x1x1 convs run as ONE GEMM whose channels-last output
rows already are the reference's flattened (x, y, z, a, c) order."""
from typing import List, Tuple

import torch
from torch import nn, Tensor

from .. import ops
from . import hip_nn


class AnchorGenerator3D(nn.Module):
    def __init__(self, sizes, aspect_ratios, is_normalized=False):
        super().__init__()
        if is_normalized:
            raise NotImplementedError("is_normalized=True is never used by run_rpn.py (normalize_aspect_ratios = False)")
        self.sizes = sizes
        self.aspect_ratios = aspect_ratios
        self.is_normalized = is_normalized
        self.aspect_ratios_unique = [ops.unique_ratio_permutations(r) for r in aspect_ratios]
        self._tables = {}

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios_unique)]

    def table(self, mesh_size, grids, device) -> "ops.AnchorTable":
        key = (tuple(mesh_size), tuple(map(tuple, grids)), str(device))
        if key not in self._tables:
            self._tables[key] = ops.AnchorTable(mesh_size, grids, self.sizes, self.aspect_ratios, device)
        return self._tables[key]

    def generate_anchors(self, scales, xyz_ratios, dtype=torch.float32, device="cpu"):
        return ops.base_anchor_table(scales, xyz_ratios).to(device=device, dtype=dtype)

    def padding_mask(self, mesh_size, grids, ori_sizes, device):
        """bool [N, T]: True for anchors whose cell lies in the un-padded part of scene n (reference anchor.py:124-152)."""
        A = self.num_anchors_per_location()[0]
        rows = []
        for o in ori_sizes:
            per_level = []
            for g in grids:
                stride = [mesh_size[i] // g[i] for i in range(3)]
                lim = [-(-int(o[i]) // stride[i]) for i in range(3)]
                mx = (torch.arange(g[0], device=device) < lim[0])[:, None, None]
                my = (torch.arange(g[1], device=device) < lim[1])[None, :, None]
                mz = (torch.arange(g[2], device=device) < lim[2])[None, None, :]
                per_level.append((mx & my & mz)[..., None].expand(g[0], g[1], g[2], A).reshape(-1))
            rows.append(torch.cat(per_level))
        return torch.stack(rows)

    def forward(self, meshes: Tensor, feature_maps: List[Tensor]):
        grids = [tuple(f.shape[-3:]) for f in feature_maps]
        tab = self.table(tuple(meshes.shape[-3:]), grids, feature_maps[0].device)
        flat = ops.anchors(tab)
        per_level = list(flat.split(tab.counts))
        non_cat = [list(per_level) for _ in range(meshes.shape[0])]
        return [flat for _ in range(meshes.shape[0])], non_cat


class RPNHead(nn.Module):
    def __init__(self, in_channels, num_anchors, conv_depth=1, rotate=False):
        super().__init__()
        convs = []
        for _ in range(conv_depth):
            convs += [nn.Conv3d(in_channels, in_channels, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
        self.conv = nn.Sequential(*convs)
        self.num_anchors = num_anchors
        self.delta_width = 8 if rotate else 6
        self.cls_logits = nn.Conv3d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.bbox_pred = nn.Conv3d(in_channels, num_anchors * self.delta_width, kernel_size=1, stride=1)
        for layer in self.modules():
            if isinstance(layer, nn.Conv3d):
                torch.nn.init.normal_(layer.weight, std=0.01)
                if layer.bias is not None:
                    torch.nn.init.constant_(layer.bias, 0)
        for m in (self.cls_logits, self.bbox_pred):      # rows of ONE fused GEMM: a trainer must keep these in the reference layout
            m.__dict__["_nrpn_fused_gemm"] = True
        used = num_anchors * (1 + self.delta_width)
        self.head_rows = ((used + 63) // 64) * 64      # padded GEMM width (128 for 13 anchors)
        self._pack = ops.PackedWeight()
        self.ragged = False        # True: the coarser levels run as one ragged launch per layer (measured: no gain)

    def forward_fused(self, feats_cl: List[Tensor]) -> List[Tensor]:
        """channels-last features -> per level fp32 [N,X,Y,Z,head_rows]: columns [0,A) logits, [A, A+A*dw) deltas."""
        n_seg = (len(feats_cl) - 1) * feats_cl[0].shape[0]
        if self.ragged and len(feats_cl) > 2 and n_seg <= 16:
            # The head shares its weights across the pyramid.  The finest level fills the chip on its own (250 tiles of 256x256 =
            # one round of the 256 CUs; adding the others would spill into a second round), so it keeps its own launches; the
            # coarser levels -- launch- and latency-bound one by one -- run as ONE launch per layer on a ragged voxel list.
            fine = hip_nn.run_modules(self.conv, feats_cl[0])
            out0 = ops.ConvFn.apply(fine, self._pack, self.head_rows, False, True, 2, self.cls_logits.weight, self.bbox_pred.weight,
                                    self.cls_logits.bias, self.bbox_pred.bias)
            x, segs = hip_nn.ragged_cat(feats_cl[1:])
            mods = list(self.conv)
            for i in range(0, len(mods), 2):
                x = hip_nn.conv3d(mods[i], x, relu=True, segs=segs)
            h = ops.ConvFn.apply(x, self._pack, self.head_rows, False, True, 2, self.cls_logits.weight, self.bbox_pred.weight,
                                 self.cls_logits.bias, self.bbox_pred.bias)
            return [out0] + hip_nn.ragged_split(h, feats_cl[1:])
        outs = []
        convs = [m for m in self.conv if isinstance(m, nn.Conv3d)]
        for f in feats_cl:
            # conv+ReLU chain private to the head: every intermediate has exactly one consumer, so each layer's ReLU backward rides
            # in the epilogue of the next layer's dgrad (ops.CHAIN_*) -- no separate relu_backward launches
            t = f
            for i, cv in enumerate(convs):
                chain = ops.CHAIN_GRAD_PREMASKED | (ops.CHAIN_MASK_INPUT_GRAD if i > 0 else 0)
                t = hip_nn.conv3d(cv, t, relu=True, chain=chain)
            # the output GEMM may mask its input gradient only when its input IS the ReLU output of a private head conv: with
            # conv_depth == 0 it reads the raw FPN feature (other consumers, no ReLU) and must hand back the full gradient
            out_chain = ops.CHAIN_MASK_INPUT_GRAD if convs else 0
            outs.append(ops.ConvFn.apply(t, self._pack, self.head_rows, (False, out_chain), True, 2, self.cls_logits.weight,
                                         self.bbox_pred.weight, self.cls_logits.bias, self.bbox_pred.bias))
        return outs

    def cone_depth(self):
        """Number of the lists S_0 .. S_depth a cone plan of this head needs (ops.ConePlan), or None when the head is not the plain
        [Conv3d k3 + ReLU] x conv_depth chain the cone evaluation is written for."""
        mods = list(self.conv)
        if len(mods) % 2:
            return None
        for i in range(0, len(mods), 2):
            cv = mods[i]
            if not (isinstance(cv, nn.Conv3d) and cv.kernel_size == (3, 3, 3) and cv.stride == (1, 1, 1) and cv.padding == (1, 1, 1)
                    and cv.in_channels == cv.out_channels and isinstance(mods[i + 1], nn.ReLU)):
                return None
        return max(len(mods) // 2 - 1, 0)

    def forward_cone(self, feats_cl: List[Tensor], plan):
        """Training only: logits [N,T] / deltas [N,T,dw] that are exact on the voxels of the sampled anchors and zero elsewhere
        (ops.ConeHeadFn: the head evaluated on the sampled-anchor cones of ``plan``)."""
        convs = [m for m in self.conv if isinstance(m, nn.Conv3d)]
        return ops.ConeHeadFn.apply(plan, self, self.num_anchors, self.delta_width, *feats_cl, *[c.weight for c in convs], *[c.bias for c in convs],
                                    self.cls_logits.weight, self.bbox_pred.weight, self.cls_logits.bias, self.bbox_pred.bias)

    def forward(self, x: List[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
        dt = x[0].dtype
        fused = self.forward_fused([hip_nn.as_ndhwc(f, dt) for f in x])
        A, dw = self.num_anchors, self.delta_width
        logits = [hip_nn.as_ncdhw(h[..., :A]) for h in fused]
        bbox = [hip_nn.as_ncdhw(h[..., A:A + A * dw]) for h in fused]
        return logits, bbox

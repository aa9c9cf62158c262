"""3D FCOS over NeRF grids on the HIP kernels, with the reference's class names, constructor arguments, state-dict keys
and forward contracts (reference nerf_rpn/model/fcos/fcos.py:17-386).

Data flow: backbone maps (channels-last) -> two towers of 4 x [3x3x3 MFMA implicit GEMM + GroupNorm(32)+ReLU kernel] ->
two fused 3x3x3 GEMMs (cls_logits [+centerness] and bbox_pred [+centerness], rows padded to 64) -> head epilogue kernel
(Scale, ReLU, stride) writing logits / regressions / centerness directly in the flattened (level, scene, voxel) order that
the loss and the post-processor consume.  Locations are index arithmetic inside the kernels."""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from .. import hip_nn
from .inference import FCOSPostProcessor
from .loss import FCOSLossComputation

HEAD_ROWS = 64      # rows of the fused final GEMMs (1 + 1 or 8 + 1 real rows, padded to one 64-row tile)


class Scale(nn.Module):
    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]))

    def forward(self, input):
        return input * self.scale


class FCOSHead(nn.Module):
    def __init__(self, in_channels, num_convs, fpn_strides, norm_reg_targets=True, centerness_on_reg=True, use_obb=False):
        super().__init__()
        self.fpn_strides, self.norm_reg_targets, self.centerness_on_reg = fpn_strides, norm_reg_targets, centerness_on_reg
        self.num_convs, self.use_obb = num_convs, use_obb

        def tower():
            mods = []
            for _ in range(num_convs):
                mods += [nn.Conv3d(in_channels, in_channels, kernel_size=3, stride=1, padding=1, bias=True), nn.GroupNorm(32, in_channels),
                         nn.ReLU()]
            return nn.Sequential(*mods)
        self.add_module("cls_tower", tower())
        self.add_module("bbox_tower", tower())
        self.cls_logits = nn.Conv3d(in_channels, 1, kernel_size=3, stride=1, padding=1)
        self.bbox_pred = nn.Conv3d(in_channels, 8 if use_obb else 6, kernel_size=3, stride=1, padding=1)
        self.centerness = nn.Conv3d(in_channels, 1, kernel_size=3, stride=1, padding=1)
        for modules in (self.cls_tower, self.bbox_tower, self.cls_logits, self.bbox_pred, self.centerness):
            for l in modules.modules():
                if isinstance(l, nn.Conv3d):
                    torch.nn.init.normal_(l.weight, std=0.01)
                    torch.nn.init.constant_(l.bias, 0)
        prior_prob = 0.01
        torch.nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_prob) / prior_prob))
        self.scales = nn.ModuleList([Scale(init_value=1.0) for _ in range(5)])
        self.__dict__["_packs"] = (ops.PackedWeight(), ops.PackedWeight())
        for m in (self.cls_logits, self.bbox_pred, self.centerness):     # rows of fused GEMMs: reference layout in a trainer's arena
            m.__dict__["_nrpn_fused_gemm"] = True

    @property
    def reg_dim(self):
        return 8 if self.use_obb else 6

    def _tower(self, tower, x):
        mods = list(tower)
        for i in range(0, len(mods), 3):
            x = hip_nn.conv3d(mods[i], x)
            x = ops.GroupNormFn.apply(x, mods[i + 1].weight, mods[i + 1].bias, mods[i + 1].num_groups, mods[i + 1].eps, True)
        return x

    def forward_flat(self, feats_cl):
        """channels-last maps -> per-level (logits [N*vox], reg [N*vox, D], centerness [N*vox])."""
        out = []
        cls_w = [self.cls_logits] + ([] if self.centerness_on_reg else [self.centerness])
        box_w = [self.bbox_pred] + ([self.centerness] if self.centerness_on_reg else [])
        pk_cls, pk_box = self.__dict__["_packs"]
        for l, f in enumerate(feats_cl):
            ct, bt = self._tower(self.cls_tower, f), self._tower(self.bbox_tower, f)
            co = ops.ConvFn.apply(ct, pk_cls, HEAD_ROWS, False, True, len(cls_w), *[m.weight for m in cls_w], *[m.bias for m in cls_w])
            bo = ops.ConvFn.apply(bt, pk_box, HEAD_ROWS, False, True, len(box_w), *[m.weight for m in box_w], *[m.bias for m in box_w])
            stride_mul = 1.0 if (self.training or not self.norm_reg_targets) else float(self.fpn_strides[l])
            out.append(ops.FcosHeadOutFn.apply(co, bo, self.scales[l].scale, stride_mul, self.norm_reg_targets, self.reg_dim,
                                               self.centerness_on_reg))
        return out

    def forward(self, x):
        """reference contract: lists of [N,1,W,L,H], [N,6|8,W,L,H], [N,1,W,L,H] (views of the flat tensors)."""
        dt = x[0].dtype
        feats = [hip_nn.as_ndhwc(f, dt) for f in x]
        logits, bbox_reg, centerness = [], [], []
        for f, (lg, rg, ct) in zip(feats, self.forward_flat(feats)):
            n, w, l, h, _ = f.shape
            logits.append(lg.view(n, w, l, h, 1).permute(0, 4, 1, 2, 3))
            bbox_reg.append(rg.view(n, w, l, h, -1).permute(0, 4, 1, 2, 3))
            centerness.append(ct.view(n, w, l, h, 1).permute(0, 4, 1, 2, 3))
        return logits, bbox_reg, centerness


class FCOSModule(nn.Module):
    def __init__(self, args, in_channels, fpn_strides, world_size=1):
        super().__init__()
        self.head = FCOSHead(in_channels, args.num_convs, fpn_strides, norm_reg_targets=args.norm_reg_targets,
                             centerness_on_reg=args.centerness_on_reg, use_obb=args.rotated_bbox)
        self.box_selector_test = FCOSPostProcessor(args.pre_nms_thresh, args.pre_nms_top_n, args.nms_thresh, args.fpn_post_nms_top_n,
                                                   args.min_size, 1, use_obb=args.rotated_bbox)
        self.loss_evaluator = FCOSLossComputation(fpn_strides, args.center_sampling_radius, args.iou_loss_type, args.norm_reg_targets,
                                                  world_size=world_size, use_obb=args.rotated_bbox,
                                                  use_additional_l1_loss=args.use_additional_l1_loss,
                                                  proj2d_loss_weight=args.proj2d_loss_weight)
        self.fpn_strides, self.world_size = fpn_strides, world_size

    def forward(self, grid_sizes, features, targets=None, objectness_output_paths=None, per_level=None, prepared=None):
        """``per_level``: the head's per-level (logits, reg, centerness) when the caller already ran it (FCOSOverNeRF with the trunk AND the
        head inside one captured HIP graph)."""
        dt = features[0].dtype
        feats = [hip_nn.as_ndhwc(f, dt) for f in features]
        n = feats[0].shape[0]
        geom = ops.FcosGeometry(n, [f.shape[1:4] for f in feats], self.fpn_strides[:len(feats)])
        if per_level is None:
            per_level = self.head.forward_flat(feats)
        logits = torch.cat([p[0] for p in per_level])
        reg = torch.cat([p[1] for p in per_level])
        ctr = torch.cat([p[2] for p in per_level])
        pad_sizes = grid_sizes if n > 1 else None              # padding masks only for batches (fcos.py:176)
        if objectness_output_paths is not None:
            self.output_objectness(geom, logits, ctr, grid_sizes, objectness_output_paths)
        if self.training:
            loss_cls, loss_reg, loss_ctr = self.loss_evaluator(geom, logits, reg, ctr, targets, pad_sizes, prepared)
            return None, None, {"loss_cls": loss_cls, "loss_reg": loss_reg, "loss_centerness": loss_ctr}
        boxes, scores = self.box_selector_test(geom, logits.detach(), reg.detach(), ctr.detach(), grid_sizes, pad_sizes)
        return boxes, scores, {}

    def compute_locations(self, features):
        """Materialised locations (reference fcos.py:221-250); the kernels never need them, kept for API parity / tests."""
        out = []
        for level, f in enumerate(features):
            w, l, h = f.size()[-3:]
            s = self.fpn_strides[level]
            g = torch.meshgrid(*[torch.arange(0, k * s, step=s, dtype=torch.float32, device=f.device) for k in (w, l, h)], indexing="ij")
            out.append(torch.stack([t.reshape(-1) for t in g], dim=1) + s // 2)
        return out

    def output_objectness(self, geom, logits, ctr, ori_sizes, output_paths):
        score = torch.sqrt(logits.detach().sigmoid() * ctr.detach().sigmoid())
        for i in range(len(ori_sizes)):
            all_levels = {}
            for level in range(geom.levels):
                lo = geom.segment_offsets[level * geom.n + i]
                d = geom.dims[level]
                w, l, h = np.ceil(np.array(ori_sizes[i]) / self.fpn_strides[level]).astype(int)
                all_levels[str(level)] = score[lo:lo + geom.counts[level]].view(*d)[:w, :l, :h].cpu().numpy()
            np.savez_compressed(output_paths[i], **all_levels)


class _TrunkAndHead:
    """backbone + FPN + the FCOS head towers as ONE callable with static shapes and no host decisions -- what graphs.GraphedBackbone captures
    for the FCOS model (round 5: the head's ~150 launches per step were the eager part that kept the Swin-S + FCOS step on the host's
    enqueue rate).  Returns the feature maps followed by the per-level (logits, reg, centerness) triples."""

    def __init__(self, backbone, head):
        self.backbone, self.head = backbone, head

    @property
    def training(self):
        return self.backbone.training and self.head.training

    @property
    def compute_dtype(self):
        return self.backbone.compute_dtype

    def parameters(self):
        yield from self.backbone.parameters()
        yield from self.head.parameters()

    def __call__(self, x):
        feats = list(self.backbone(x))
        cl = [hip_nn.as_ndhwc(f, feats[0].dtype) for f in feats]
        per = self.head.forward_flat(cl)
        return (*feats, *[t for triple in per for t in triple])


class FCOSOverNeRF(nn.Module):
    """Backbone + FCOS head (reference fcos.py:289-386); ``compute_dtype`` as in NeRFRegionProposalNetwork."""

    def __init__(self, args, backbone, fpn_strides, world_size=1, compute_dtype=torch.float32):
        if not hasattr(backbone, "out_channels"):
            raise ValueError("backbone should contain an attribute out_channels specifying the number of output channels "
                             "(assumed to be the same for all the levels)")
        super().__init__()
        self.args, self.world_size = args, world_size
        self.backbone = backbone
        self.fcos_module = FCOSModule(args, backbone.out_channels, fpn_strides, world_size=world_size)
        from ... import graphs as _graphs
        self.use_graph = _graphs.ENABLED[0]     # training: backbone + FPN forward / backward as two captured HIP graphs (needs an engine.FlatTrainer)
        self._trunk = None
        self._prep_stream = None          # side stream of the target preparation (forward)
        self.bf16x3 = False
        self.set_compute_dtype(compute_dtype)

    def set_compute_dtype(self, dtype):
        if isinstance(dtype, str):       # "bf16x3": see NeRFRegionProposalNetwork.set_compute_dtype
            if dtype not in ("bf16x3", "fp32", "bf16"):
                raise ValueError("compute_dtype must be torch.float32, torch.bfloat16 or 'bf16x3'")
            self.bf16x3 = dtype == "bf16x3"
            dtype = torch.bfloat16 if dtype == "bf16" else torch.float32
        else:
            self.bf16x3 = False
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32, torch.bfloat16 or 'bf16x3'")
        self.compute_dtype = dtype
        self.backbone.compute_dtype = dtype
        return self

    def transform(self, meshes):
        tgt = np.max([m.shape for m in meshes], axis=0)
        return [F.pad(m, (0, int(tgt[-1] - m.shape[-1]), 0, int(tgt[-2] - m.shape[-2]), 0, int(tgt[-3] - m.shape[-3])), mode="constant",
                      value=0) for m in meshes]

    def forward(self, meshes, targets=None, objectness_output_paths=None):
        prev = ops.SPLIT3[0]        # bf16x3 is a property of this model: see NeRFRegionProposalNetwork.forward
        ops.SPLIT3[0] = prev or self.bf16x3
        try:
            return self._forward_impl(meshes, targets, objectness_output_paths)
        finally:
            ops.SPLIT3[0] = prev

    def _forward_impl(self, meshes, targets=None, objectness_output_paths=None):
        if self.training:
            if targets is None:
                torch._assert(False, "targets should not be none when in training mode")
            width = 7 if self.args.rotated_bbox else 6
            for boxes in targets:
                if isinstance(boxes, torch.Tensor):
                    torch._assert(len(boxes.shape) == 2 and boxes.shape[-1] == width,
                                  f"Expected target boxes to be a tensor of shape [N, {width}], got {boxes.shape}.")
                else:
                    torch._assert(False, f"Expected target boxes to be of type Tensor, got {type(boxes)}.")
        sizes = []
        for mesh in meshes:
            val = mesh.shape[-3:]
            torch._assert(len(val) == 3, f"expecting the last three dimensions of the Tensor to be W, L and H instead got {mesh.shape[-3:]}")
            sizes.append((int(val[0]), int(val[1]), int(val[2])))
        meshes = list(meshes)
        if len(meshes) > 1:
            meshes = self.transform(meshes)
        stacked = ops.stack_scenes(meshes)
        prepared = None
        if self.training and stacked.is_cuda and hasattr(self.backbone, "feature_grids"):
            # Target assignment, the positive-location list and the loss normalisers hold every host synchronisation of an FCOS step and
            # depend on the ground truth only: they are issued BEFORE the backbone, on their own stream when the boxes arrive as host
            # tensors (what the reference's loader yields), so the host never waits for the head's outputs in the middle of the step
            # (round 5: the step was forward + [sync] + loss / backward enqueue in series).
            dev = stacked.device
            n = int(stacked.shape[0])
            grids = [tuple(int(v) for v in g) for g in self.backbone.feature_grids(tuple(int(v) for v in stacked.shape[-3:]))]
            fm = self.fcos_module
            geom = ops.FcosGeometry(n, grids, fm.fpn_strides[:len(grids)])
            pad_sizes = sizes if n > 1 else None
            main = torch.cuda.current_stream(dev)
            if self._prep_stream is None:
                self._prep_stream = torch.cuda.Stream(device=dev)
            if any(t.is_cuda for t in targets):
                self._prep_stream.wait_stream(main)           # device boxes may still have producers on the main stream
            with torch.cuda.stream(self._prep_stream):
                dev_targets = [t.to(dev, non_blocking=True) for t in targets]
                prepared = fm.loss_evaluator.prepare(geom, dev_targets, pad_sizes, dev)
            main.wait_stream(self._prep_stream)
            for v in list(prepared.values()) + dev_targets:
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(main)
            targets = dev_targets
        per_level = None
        if self.use_graph and self.training and stacked.is_cuda and torch.is_grad_enabled():
            if self._trunk is None:
                from ...graphs import GraphedBackbone
                self._trunk = GraphedBackbone(_TrunkAndHead(self.backbone, self.fcos_module.head))
            outs = list(self._trunk(stacked))           # backbone + FPN + head towers as captured HIP graphs (graphs.py); eager until captured
            L = len(outs) // 4
            features = outs[:L]
            per_level = [tuple(outs[L + 3 * l:L + 3 * l + 3]) for l in range(L)]
        else:
            features = list(self.backbone(stacked))
        boxes, scores, losses = self.fcos_module(sizes, features, targets, objectness_output_paths, per_level, prepared)
        return boxes, losses, scores

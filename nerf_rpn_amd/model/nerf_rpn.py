"""Top-level detector with the reference's class name, constructor and ``forward`` contract
(reference nerf_rpn/model/nerf_rpn.py:21-217), running on the HIP kernels.

Extra keyword (absorbed by ``**kwargs`` in the reference, so call sites stay source-compatible):
  compute_dtype = torch.float32 (bit-for-bit-comparable parity path, default) | torch.bfloat16 (throughput path)."""
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from .anchor import AnchorGenerator3D, RPNHead
from .rpn import RegionProposalNetwork


def _default_anchorgen():
    sizes = ((8,), (16,), (32,), (64,),)
    ratios = (((1., 1., 1.), (1., 1., 2.), (1., 2., 2.), (1., 1., 3.), (1., 3., 3.)),) * len(sizes)
    return AnchorGenerator3D(sizes, ratios)


class NeRFRegionProposalNetwork(nn.Module):
    def __init__(self, backbone, rpn_anchor_generator=None, rpn_head=None, rpn_pre_nms_top_n_train=2000, rpn_pre_nms_top_n_test=1000,
                 rpn_post_nms_top_n_train=2000, rpn_post_nms_top_n_test=1000, rpn_nms_thresh=0.7, rpn_fg_iou_thresh=0.7,
                 rpn_bg_iou_thresh=0.3, rpn_batch_size_per_image=256, rpn_positive_fraction=0.5, rpn_score_thresh=0.0,
                 iou_batch_size=16, rotated_bbox=False, reg_loss_type="smooth_l1", **kwargs):
        if not hasattr(backbone, "out_channels"):
            raise ValueError("backbone should contain an attribute out_channels specifying the number of output channels "
                             "(assumed to be the same for all the levels)")
        if not isinstance(rpn_anchor_generator, (AnchorGenerator3D, type(None))):
            raise TypeError(f"rpn_anchor_generator should be of type AnchorGenerator or None instead of {type(rpn_anchor_generator)}")
        if rpn_anchor_generator is None:
            rpn_anchor_generator = _default_anchorgen()
        if rpn_head is None:
            # NB the reference passes rotated_bbox into the conv_depth slot here (nerf_rpn.py:100); kept source-compatible
            rpn_head = RPNHead(backbone.out_channels, rpn_anchor_generator.num_anchors_per_location()[0], rotated_bbox)
        super().__init__()
        self.backbone = backbone
        self.rpn = RegionProposalNetwork(
            rpn_anchor_generator, rpn_head, rpn_fg_iou_thresh, rpn_bg_iou_thresh, rpn_batch_size_per_image, rpn_positive_fraction,
            dict(training=rpn_pre_nms_top_n_train, testing=rpn_pre_nms_top_n_test),
            dict(training=rpn_post_nms_top_n_train, testing=rpn_post_nms_top_n_test), rpn_nms_thresh,
            score_thresh=rpn_score_thresh, iou_batch_size=iou_batch_size, rotated_bbox=rotated_bbox, reg_loss_type=reg_loss_type)
        self._prep_stream = None          # side stream of the target preparation when the ground truth arrives as host tensors (forward)
        from .. import graphs as _graphs
        self.use_graph = _graphs.ENABLED[0]     # training: backbone + FPN forward / backward as two captured HIP graphs (needs an engine.FlatTrainer);
                                                # "fwd": only the forward is captured, the backward stays eager (graphs.GraphedBackbone)
        self._trunk = None
        self.bf16x3 = False
        self.set_compute_dtype(kwargs.get("compute_dtype", torch.float32))

    def set_compute_dtype(self, dtype):
        """torch.float32 (exact fp32 MFMA chains: the parity mode), torch.bfloat16 (throughput) or "bf16x3": fp32 activations / weights /
        gradients with the 3x3x3 convolutions evaluated as three bf16 MFMA products of split operands (fp32 results to fp32 accumulation
        error at a fraction of the fp32 MFMA time; a property of this model -- ops.SPLIT3 / NRPN_BF16X3=1 is the process-wide form)."""
        if isinstance(dtype, str):
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
        self.rpn.compute_dtype = dtype
        return self

    def transform(self, meshes, targets=None):
        if len(meshes) > 1:
            tgt = np.max([m.shape for m in meshes], axis=0)
            meshes = [F.pad(m, (0, int(tgt[-1] - m.shape[-1]), 0, int(tgt[-2] - m.shape[-2]), 0, int(tgt[-3] - m.shape[-3])),
                            mode="constant", value=0) for m in meshes]
        return meshes, targets

    @staticmethod
    def _degenerate(boxes):
        return boxes[:, 3:] <= boxes[:, :3] if boxes.shape[1] == 6 else boxes[:, 3:6] <= 0

    def check_bbox_degeneration(self, targets):
        if targets is None:
            return
        for target_idx, boxes in enumerate(targets):
            bad = self._degenerate(boxes)
            if bad.any():
                bb = boxes[torch.where(bad.any(dim=1))[0][0]].tolist()
                torch._assert(False, "All bounding boxes should have positive height, width and depth."
                                     f" Found invalid box {bb} for target at index {target_idx}.")

    def forward(self, meshes, targets=None, objectness_output_paths=None):
        # bf16x3 is a property of THIS model: the process-wide switch of the conv functions (ops.SPLIT3) is raised for the duration of its
        # forward pass only (backward passes follow what their forward recorded), so an fp32 and a bf16x3 model can live in one process
        prev = ops.SPLIT3[0]
        ops.SPLIT3[0] = prev or self.bf16x3
        try:
            return self._forward_impl(meshes, targets, objectness_output_paths)
        finally:
            ops.SPLIT3[0] = prev

    def _forward_impl(self, meshes, targets=None, objectness_output_paths=None):
        if self.training:
            if targets is None:
                torch._assert(False, "targets should not be none when in training mode")
            for boxes in targets:
                if isinstance(boxes, torch.Tensor):
                    torch._assert(len(boxes.shape) == 2 and boxes.shape[-1] in (6, 7),
                                  f"Expected target boxes to be a tensor of shape [N, 6], got {boxes.shape}.")
                else:
                    torch._assert(False, f"Expected target boxes to be of type Tensor, got {type(boxes)}.")
        original_mesh_sizes: List[Tuple[int, int, int]] = []
        for mesh in meshes:
            val = mesh.shape[-3:]
            torch._assert(len(val) == 3, f"expecting the last three dimensions of the Tensor to be W, H and D instead got {mesh.shape[-3:]}")
            original_mesh_sizes.append((int(val[0]), int(val[1]), int(val[2])))
        meshes, targets = self.transform(list(meshes), targets)
        early = self.training and hasattr(self.backbone, "feature_grids")
        if not early:
            self.check_bbox_degeneration(targets)
        mesh_tensors = ops.stack_scenes(meshes)
        prepared = None
        if early:
            # target assignment + sampling are issued before the backbone and hold the ONE host read-back of a training step (the sampled
            # counts); the degenerate-box check rides on that copy as device flags instead of synchronising on its own.  The rest of
            # the step is then enqueued without a synchronisation (see RegionProposalNetwork.prepare_targets)
            size = tuple(int(v) for v in mesh_tensors.shape[-3:])
            grids = [tuple(g) for g in self.backbone.feature_grids(size)]
            dev = mesh_tensors.device
            if dev.type == "cuda" and all(not b.is_cuda for b in targets):
                # Ground truth handed over as HOST tensors (what the reference's loader yields): nothing of the target preparation
                # depends on earlier GPU work, so it runs on its own stream -- upload, matcher, sampler and the read-back of the sampled
                # counts -- and the host never waits for the previous step's backward still executing on the main stream.
                self.check_bbox_degeneration(targets)         # on the host copies: no device round trip
                main = torch.cuda.current_stream(dev)
                if self._prep_stream is None:
                    self._prep_stream = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(self._prep_stream):
                    targets = [b.to(dev, non_blocking=True) for b in targets]
                    prepared = self.rpn.prepare_targets(size, grids, targets, original_mesh_sizes, dev)
                main.wait_stream(self._prep_stream)
                for v in list(prepared.values()) + [targets]:       # produced on the side stream, consumed (and later freed) on the main one
                    if isinstance(v, ops.ConePlan):
                        v = v.tensors()
                    for t in (v if isinstance(v, (list, tuple)) else [v]):
                        if isinstance(t, torch.Tensor) and t.is_cuda:
                            t.record_stream(main)
            else:
                flags = [self._degenerate(b).any() for b in targets if b.is_cuda and b.numel()]
                prepared = self.rpn.prepare_targets(size, grids, targets, original_mesh_sizes, dev, flags)
                if prepared["flags"] is None or any(prepared["flags"]) or len(prepared["flags"]) != len(targets):
                    self.check_bbox_degeneration(targets)         # raises with the offending box (or covers what the flags did not)
        if self.use_graph and self.training and mesh_tensors.is_cuda and torch.is_grad_enabled():
            if self._trunk is None:
                from ..graphs import GraphedBackbone
                self._trunk = GraphedBackbone(self.backbone, backward="eager" if self.use_graph == "fwd" else "graph")
            features = list(self._trunk(mesh_tensors))       # backbone + FPN as captured HIP graphs (graphs.py); eager until captured
        else:
            features = list(self.backbone(mesh_tensors))
        proposals, level_index, proposal_losses, scores = self.rpn(mesh_tensors, features, original_mesh_sizes, targets,
                                                                   objectness_output_paths, prepared)
        return [features, proposals, level_index], proposal_losses, scores

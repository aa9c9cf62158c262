"""Region proposal network on the HIP kernels.  Constructor and ``forward`` contract mirror reference
nerf_rpn/model/rpn.py:167-536; the body is re-designed for the device:

  eval : head GEMM -> flatten kernel -> per-level radix top-k -> decode ONLY the <= 4 x pre_nms_top_n candidates (anchors
         computed from their index) -> one filter/compaction kernel -> bitmask NMS (IoU tiles + LDS scan) -> final sort;
         a single host read-back per scene (the proposal count).
  train: fused IoU+matcher over the 950k anchors (no [G, T] matrix is materialised), device-side sampling, targets
         encoded only for the sampled positives, BCE + smooth-L1 with fused backward that writes straight into the head
         gradient rows.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn, Tensor

from .. import ops
from . import hip_nn
from .anchor import AnchorGenerator3D
from .coder import AABBCoder, MidpointOffsetCoder
from .coder.misc import obb2hbb_3d, obb2points_3d
from .rotated_iou.oriented_iou_loss import cal_diou_3d, cal_giou_3d, cal_iou_3d
from .utils import BalancedPositiveNegativeSampler, Matcher


def permute_and_flatten(layer: Tensor, N: int, A: int, C: int, W: int, H: int, D: int) -> Tensor:
    """(N, A*C, W, H, D) -> (N, W*H*D*A, C); free when ``layer`` is a channels-last-backed view."""
    return layer.view(N, -1, C, W, H, D).permute(0, 3, 4, 5, 1, 2).reshape(N, -1, C)


_VIEW_CACHE = {}


def _view_stack(res, device):
    """(M [4,4,4] world->camera matrices, K [3,3] intrinsics) on ``device``; host numpy + upload happen once per (res, device)
    instead of every step (the reference rebuilds and uploads them in every compute_loss call, rpn.py:421-453)."""
    key = (float(res), str(device))
    if key not in _VIEW_CACHE:
        K = torch.tensor([[600., 0., 320.], [0., 600., 240.], [0., 0., 1.]], device=device)
        _VIEW_CACHE[key] = (torch.stack(_view_matrices(res, device)), K)
    return _VIEW_CACHE[key]


def _view_matrices(res, device):
    """Four world->camera matrices looking at the grid centre (reference get_w2cs, rpn.py:76-83)."""
    ctr = np.array([res / 2] * 3)
    mats = []
    for p in np.array([[res, res, res], [res, -res, res], [-res, res, res], [-res, -res, res]]) + ctr:
        def unit(v):
            n = np.linalg.norm(v)
            return v / (n if n != 0 else 1)
        zax = unit(p - ctr)
        xax = unit(np.cross(np.array([0, 0, 1]), zax))
        yax = unit(np.cross(zax, xax))
        c2w = np.eye(4)
        c2w[:3, :3] = np.stack([xax, yax, zax], axis=1)
        c2w[:3, 3] = p
        mats.append(torch.tensor(np.linalg.inv(c2w), dtype=torch.float32, device=device))
    return mats


class RotatedIOULoss(nn.Module):
    def __init__(self, loss_type="iou"):
        super().__init__()
        self.loss_type = loss_type

    def forward(self, pred, target, weight=None):
        if pred.is_cuda and not target.requires_grad and self.loss_type in ("iou", "linear_iou", "giou", "diou"):
            # one fused kernel: loss and d loss / d pred of every pair (ops.RotatedIoULossFn); the torch formulation below remains for
            # the case where the target also needs a gradient
            losses = ops.rotated_iou_loss(pred, target, self.loss_type)[0]
            if weight is not None and weight.sum() > 0:
                return (losses * weight).sum()
            assert losses.numel() != 0
            return losses.sum()
        p, t = pred.unsqueeze(0), target.unsqueeze(0)
        if self.loss_type in ('iou', 'linear_iou'):
            ious, _, _, _, unions = cal_iou_3d(p, t, verbose=True)
            ious = (ious * unions + 1.0) / (unions + 1.0)
            losses = -torch.log(ious) if self.loss_type == 'iou' else 1 - ious
        elif self.loss_type == 'giou':
            losses, _, _ = cal_giou_3d(p, t)
        elif self.loss_type == 'diou':
            losses, _ = cal_diou_3d(p, t)
        else:
            raise NotImplementedError
        if weight is not None and weight.sum() > 0:
            return (losses * weight).sum()
        assert losses.numel() != 0
        return losses.sum()


class RegionProposalNetwork(nn.Module):
    def __init__(self, anchor_generator: AnchorGenerator3D, head: nn.Module, fg_iou_thresh: float, bg_iou_thresh: float,
                 batch_size_per_mesh: int, positive_fraction: float, pre_nms_top_n: Dict[str, int], post_nms_top_n: Dict[str, int],
                 nms_thresh: float, score_thresh: float = 0.0, iou_batch_size: int = 16, rotated_bbox: bool = False,
                 reg_loss_type: str = "smooth_l1"):
        super().__init__()
        self.anchor_generator = anchor_generator
        self.head = head
        self.rotate = rotated_bbox
        self.box_coder = AABBCoder() if not rotated_bbox else MidpointOffsetCoder()
        self.num_bbox_digits = 6 if not rotated_bbox else 7
        self.num_delta_digits = 6 if not rotated_bbox else 8
        self.iou_batch_size = iou_batch_size
        self.proposal_matcher = Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=True)
        self.fg_bg_sampler = BalancedPositiveNegativeSampler(batch_size_per_mesh, positive_fraction)
        self._pre_nms_top_n = pre_nms_top_n
        self._post_nms_top_n = post_nms_top_n
        self.nms_thresh = nms_thresh
        self.score_thresh = score_thresh
        if not rotated_bbox and reg_loss_type != "smooth_l1":
            # the reference builds RotatedIOULoss regardless and crashes in its 7-column box maths on 6-column AABBs (rpn.py:133-164)
            raise ValueError(f"reg_loss_type={reg_loss_type!r} needs --rotated_bbox (axis-aligned boxes train with smooth_l1 only)")
        self.reg_loss_type = reg_loss_type
        self.rotated_iou_loss = RotatedIOULoss(reg_loss_type) if rotated_bbox and reg_loss_type != "smooth_l1" else None
        self.min_size = 1e-3
        # extras (not in the reference)
        self.fix_obb_clip = False            # True = drop scores/levels together with out-of-grid OBBs (quirk B3 fixed)
        self.loss_2d_requires_grad = True    # trainers set False when reg_loss_weight_2d == 0 (value is still reported)
        self.sampler_hook = None             # tests: callable(labels_list) -> (pos_idx, neg_idx) over the flat batch
        self.record_stages = False           # tests / diagnosis: keep references to the eval stage tensors of the last forward in last_aux
        self.use_cone = True                 # training: evaluate the head on the sampled-anchor cones only (ops.ConeHeadFn); False = dense head
        self.compute_dtype = torch.float32
        self.last_aux = {}
        self.last_cone = None

    def pre_nms_top_n(self) -> int:
        return self._pre_nms_top_n["training" if self.training else "testing"]

    def post_nms_top_n(self) -> int:
        return self._post_nms_top_n["training" if self.training else "testing"]

    # -------------------------------------------------------------------------------------------------------- eval
    def filter_proposals(self, table, logits: Tensor, deltas: Tensor, mesh_shapes, padding_masks: Optional[Tensor]):
        """logits [N,T], deltas [N,T,dw] -> per-scene (boxes, scores, levels); reference rpn.py:292-370."""
        logits = logits.detach()
        if padding_masks is not None:
            logits = logits.masked_fill(~padding_masks, float("-inf"))
        k = self.pre_nms_top_n()
        L0 = len(table.counts)
        if k * L0 > 16384 or self.post_nms_top_n() > 16384:
            # the candidate sort / NMS / selection kernels work on at most 16384 rows (one workgroup's LDS); the reference has no
            # such limit, so refuse loudly instead of silently clamping
            raise ValueError(f"rpn_pre_nms_top_n ({k}) x pyramid levels ({L0}) and rpn_post_nms_top_n ({self.post_nms_top_n()}) must "
                             "each stay <= 16384 on the HIP path")
        L = len(table.counts)
        dev = logits.device
        slot_level = torch.arange(L, dtype=torch.int32, device=dev).repeat_interleave(k).contiguous()
        boxes_out, scores_out, levels_out = [], [], []
        pending, stages = [], []
        for n in range(logits.shape[0]):
            idx, val = ops.segmented_topk(logits[n], table.offsets, k)
            cand = idx.reshape(-1)
            valid = (cand >= 0).to(torch.uint8)
            boxes = ops.decode_boxes(table, deltas[n].detach(), cand.long(), int(self.rotate))
            fb, fs, fl, cnt = ops.filter_candidates(boxes, val.reshape(-1).contiguous(), slot_level, valid, mesh_shapes[n], self.min_size,
                                                    self.score_thresh, self.fix_obb_clip)
            keep = ops.nms3d_sorted(fb, fl, self.nms_thresh, cnt)
            pending.append(ops.select_kept(fb, fs, fl, keep, cnt, self.post_nms_top_n()))
            stages.append(dict(cand_boxes=boxes, cand_valid=valid, cand_level=slot_level, cand_logits=val.reshape(-1), nms_boxes=fb, nms_levels=fl, nms_count=cnt, nms_keep=keep))
        # parity tests look at the decisions behind a proposal list (record_stages = True); production eval keeps no references to the
        # per-scene candidate / NMS tensors beyond this call
        self.last_aux = dict(stages=stages) if self.record_stages else {}
        for ob, os_, ol, oc in pending:
            m = int(oc.item())   # the one device->host read-back per scene
            boxes_out.append(ob[:m])
            scores_out.append(os_[:m])
            levels_out.append(ol[:m])
        return boxes_out, scores_out, levels_out

    # -------------------------------------------------------------------------------------------------------- train
    def assign_targets_to_anchors(self, table, targets: List[Tensor], ori_sizes, padding_masks: Optional[Tensor] = None):
        labels, matched = [], []
        dev = table.words.device
        for i, gt in enumerate(targets):
            if gt.numel() == 0:
                lab = torch.zeros(table.total, dtype=torch.float32, device=dev)
                if padding_masks is not None:          # anchors in the zero padding of a batched scene are ignored here too (rpn.py:281-283)
                    lab = lab.masked_fill(~padding_masks[i], -1.0)
                labels.append(lab)
                matched.append(torch.zeros(table.total, dtype=torch.int32, device=dev))
                continue
            gt_aabb = obb2hbb_3d(gt) if gt.size(1) == 7 else gt
            lab, m = ops.match_anchors(table, gt_aabb, self.proposal_matcher.high_threshold, self.proposal_matcher.low_threshold,
                                       ori_sizes[i] if ori_sizes is not None else None)
            labels.append(lab)
            matched.append(m)
        return labels, matched

    def _cone_depth(self, n, grids, device):
        """Depth of the cone plan of a training step (ops.ConePlan), or None when the head runs densely: switched off, CPU tensors, a
        head that is not the plain conv chain, more (level, scene) segments than the ragged conv kernels take, or a channel row that is
        not a whole number of 128-byte K-steps."""
        if not (ops.CONE_ENABLED[0] and self.use_cone) or torch.device(device).type != "cuda" or len(grids) * n > 16:
            return None
        depth = self.head.cone_depth() if hasattr(self.head, "cone_depth") else None
        if depth is None:
            return None
        es = 2 if self.compute_dtype == torch.bfloat16 else 4
        if (self.head.cls_logits.in_channels * es) % 128 or (self.head.head_rows * es) % 128:
            return None
        return depth

    def prepare_targets(self, mesh_size, grids, targets: List[Tensor], original_mesh_sizes, device, pending_flags=None):
        """Everything of the training loss that does not depend on the network output: anchor table, IoU + matcher labels, the
        sampled positives / negatives, their matched ground truth, regression targets and anchors.  The model calls this BEFORE the
        backbone is enqueued: the sampler's host read-backs (torch.where / randperm sizes) then happen while the GPU has nothing
        queued, and the whole forward + loss + backward is issued without a host synchronisation (the reference samples after the
        head, rpn.py:506-527, which drains the launch queue in the middle of every step)."""
        n = len(targets)
        table = self.anchor_generator.table(mesh_size, grids, device)
        pad = self.anchor_generator.padding_mask(mesh_size, grids, original_mesh_sizes, device) if n > 1 else None
        labels, matched = self.assign_targets_to_anchors(table, targets, original_mesh_sizes if n > 1 else None, pad)
        T = table.total
        flags = None
        # Sampled-anchor cones: the loss reads the head at the sampled anchors only, so the head runs on their receptive-field cones
        # (ops.ConeHeadFn); the voxel lists are built right behind the sampler kernels and their sizes ride on the sampler's read-back.
        depth = self._cone_depth(n, grids, device)
        cone = ops.ConePlan(grids, n, depth, device) if depth is not None else None
        if self.sampler_hook is not None:
            pos, neg = self.sampler_hook(labels)
            if cone is not None:
                ops.cone_from_indices(cone, pos.to(device), neg.to(device), table)
        else:
            hook = None if cone is None else (lambda op_, on_, cnt_: (ops.cone_build(cone, op_, on_, cnt_, table), cone.finish))
            pairs, flags = self.fg_bg_sampler.sample_batch(labels, pending_flags, hook)     # the one host read-back of a training step
            if n == 1:
                pos, neg = pairs[0]
            else:
                pos, neg = torch.cat([p + i * T for i, (p, _) in enumerate(pairs)]), torch.cat([q + i * T for i, (_, q) in enumerate(pairs)])
        pos, neg = pos.to(device), neg.to(device)
        if self.sampler_hook is not None or n > 1:      # sample_indices already returns ascending indices per scene
            pos, neg = pos.sort()[0], neg.sort()[0]
        pos, neg = pos.contiguous(), neg.contiguous()
        if n > 1:
            scene = torch.div(pos, T, rounding_mode="floor")   # non-decreasing because pos is sorted
            local = (pos - scene * T).contiguous()
            gt_rows = []
            for i, gt in enumerate(targets):
                sel = scene == i
                if gt.numel() == 0:
                    gt_rows.append(torch.zeros((int(sel.sum()), self.num_bbox_digits), dtype=torch.float32, device=device))
                else:
                    gt_rows.append(gt.float()[matched[i][local[sel]].long()])
            matched_gt = torch.cat(gt_rows).contiguous()
        else:
            local = pos
            gt = targets[0]
            matched_gt = (gt.float()[matched[0][local].long()] if gt.numel()
                          else torch.zeros((pos.numel(), self.num_bbox_digits), dtype=torch.float32, device=device)).contiguous()
        reg_targets = ops.encode_boxes(table, matched_gt, local, int(self.rotate))
        anchors_pos = ops.anchors(table, local)
        return dict(mesh_size=tuple(mesh_size), grids=[tuple(g) for g in grids], table=table, pad=pad, labels=labels, matched=matched,
                    flags=flags, pos=pos, neg=neg, matched_gt=matched_gt, reg_targets=reg_targets, anchors_pos=anchors_pos, cone=cone)

    def compute_loss(self, prep, logits, deltas, max_mesh_dim):
        """reference rpn.py:372-456 on the sampled rows only."""
        pos, neg, matched_gt = prep["pos"], prep["neg"], prep["matched_gt"]
        flat_logits, flat_deltas = logits.reshape(-1), deltas.reshape(-1, self.num_delta_digits)
        loss_obj, loss_reg_l1 = ops.SampledLossFn.apply(flat_logits, flat_deltas, prep["reg_targets"], pos, neg, 1.0 / 9)
        need_box_grad = self.rotated_iou_loss is not None or self.loss_2d_requires_grad
        if need_box_grad:
            pred_pos = self.box_coder.decode_single_diff(flat_deltas[pos], prep["anchors_pos"])
        else:
            pred_pos = self.box_coder.decode_single(flat_deltas[pos], prep["anchors_pos"])
        if self.rotated_iou_loss is not None:
            loss_reg = self.rotated_iou_loss(pred_pos, matched_gt) / (pos.numel() + neg.numel())
        else:
            loss_reg = loss_reg_l1
        with torch.set_grad_enabled(self.loss_2d_requires_grad and torch.is_grad_enabled()):
            loss_2d = self._projection_loss(pred_pos if self.loss_2d_requires_grad else pred_pos.detach(), matched_gt, max_mesh_dim)
        self.last_aux = dict(pos=pos, neg=neg, labels=prep["labels"])
        return loss_obj, loss_reg, loss_2d

    def _projection_loss(self, pred, target, max_mesh_dim):
        """2-D projection smooth-L1 over 4 cameras (reference rpn.py:37-102, 421-453).  Same arithmetic per element -- camera =
        M @ [x y z 1]^T, picture = K @ camera[:3], u,v = picture[:2] / picture[2] -- batched over the 4 cameras and over
        prediction + target points (2 batched matmuls + 1 divide instead of 16 matmuls, 8 divides and 10 concatenations)."""
        M, K = _view_stack(max_mesh_dim, pred.device)
        if pred.is_cuda and not (pred.requires_grad and torch.is_grad_enabled()):
            return ops.projection_loss(pred, target, M, K, 1.0 / 9, float(max_mesh_dim))    # one launch instead of ~45
        if target.size(1) == 6:
            p = torch.cat([pred[:, :3], pred[:, 3:]], dim=0)
            t = torch.cat([target[:, :3], target[:, 3:]], dim=0)
        else:
            p, t = obb2points_3d(pred), obb2points_3d(target)
        npts = p.shape[0]
        src = torch.cat([p.float(), t.float()], dim=0)
        src = torch.cat([src, torch.ones(src.shape[0], 1, device=src.device)], dim=1)       # [2P, 4]
        cam = torch.matmul(M, src.t())                                                      # [4, 4, 2P]
        pic = torch.matmul(K, cam[:, :3])                                                   # [4, 3, 2P]
        uv = pic[:, :2] / pic[:, 2:3]                                                       # [4, 2, 2P]
        return F.smooth_l1_loss(uv[:, :, :npts], uv[:, :, npts:], beta=1 / 9, reduction="sum") / pred.shape[0] / max_mesh_dim

    # -------------------------------------------------------------------------------------------------------- forward
    def forward(self, meshes: Tensor, features: List[Tensor], original_mesh_sizes, targets: Optional[List[Tensor]] = None,
                objectness_output_paths=None, prepared=None):
        dt = features[0].dtype
        feats_cl = [hip_nn.as_ndhwc(f, dt) for f in features]
        n = meshes.shape[0]
        mesh_size = tuple(int(v) for v in meshes.shape[-3:])
        grids = [tuple(int(v) for v in f.shape[1:4]) for f in feats_cl]
        A, dw = self.head.num_anchors, self.num_delta_digits
        cone = prepared.get("cone") if (self.training and prepared is not None and objectness_output_paths is None
                                        and prepared["grids"] == grids and prepared["mesh_size"] == mesh_size) else None
        self.last_cone = None
        if cone is not None:
            # training: only the sampled anchors' logits / deltas are read below, so the head is evaluated on their cones (zeros elsewhere)
            logits, deltas = self.head.forward_cone(feats_cl, cone)
            self.last_cone = dict(rows=list(cone.counts), total_voxels=cone.total)       # sizes only (bench / logging)
        else:
            heads = self.head.forward_fused(feats_cl)
            if objectness_output_paths is not None:
                self.output_objectness([hip_nn.as_ncdhw(h[..., :A]) for h in heads], original_mesh_sizes, objectness_output_paths)
            logits, deltas = ops.FlattenHeadFn.apply(A, dw, dt, *[h.reshape(n, -1, h.shape[-1]) for h in heads])
        boxes = scores = level_indexes = None
        losses = {}
        if not self.training:
            table = self.anchor_generator.table(mesh_size, grids, feats_cl[0].device)
            pad = self.anchor_generator.padding_mask(mesh_size, grids, original_mesh_sizes, logits.device) if n > 1 else None
            boxes, scores, level_indexes = self.filter_proposals(table, logits, deltas, [mesh_size] * n, pad)
            if self.record_stages:       # parity tests hold the raw head outputs of every anchor to the reference's (north_star: 1e-4)
                self.last_aux["logits"], self.last_aux["deltas"] = logits.detach(), deltas.detach()
        else:
            if targets is None:
                raise ValueError("targets should not be None")
            if prepared is None or prepared["grids"] != grids or prepared["mesh_size"] != mesh_size:
                prepared = self.prepare_targets(mesh_size, grids, targets, original_mesh_sizes, logits.device)
            lo, lr, l2 = self.compute_loss(prepared, logits, deltas, max(mesh_size))
            losses = {"loss_objectness": lo, "loss_rpn_box_reg": lr, "loss_rpn_box_reg_2d": l2}
        return boxes, level_indexes, losses, scores

    def output_objectness(self, objectness, ori_sizes, output_paths):
        for i in range(len(ori_sizes)):
            levels = {}
            for level, ob in enumerate(objectness):
                score = ob[i].float().max(dim=0)[0]
                w, l, h = np.ceil(np.array(ori_sizes[i]) / 2 ** (level + 2)).astype(int)
                levels[str(level)] = score[:w, :l, :h].cpu().numpy()
            np.savez_compressed(output_paths[i], **levels)

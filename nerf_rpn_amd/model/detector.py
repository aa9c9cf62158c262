"""Second-stage objectness / refinement network on the HIP kernels, with the reference's class names and call contracts
(nerf_rpn/model/detector.py:12-627): ``ProposalTargetLayer`` (RoI <-> ground-truth assignment and fg/bg sampling), ``ROIPool`` (per-RoI
feature extraction from the pyramid), ``RCNN`` (classification + box-regression head), ``Classification_Model`` (the composition).

What runs where:
  * RoI <-> GT IoU: one fused IoU-matrix launch per scene (``ops.iou3d_matrix``, AABB or rotated);
  * RoI features: rotated 3D RoIAlign on channels-last pyramid levels (``csrc/roialign.hip``; reference op rotated_roi_3d);
  * the head's 3x3x3 convs: the MFMA implicit-GEMM conv kernels; the two Linear layers: the same GEMM with one tap.

``ROIPool`` modes:
  * ``use_cuda=False`` (the reference CLI's default): the reference's pooling WITHOUT the op as HIP kernels (csrc/roipool.hip, round 5; one
    launch per pyramid level over all RoIs of a scene, forward and backward) -- integer crops + adaptive max-pool for AABBs, the rotated
    8-corner gather + max-pool / trilinear resize for OBBs, including its in-place enlargement of the caller's OBB RoIs and its AABB
    "enlargement" by (1 + e) / 2; held to outputs of the reference itself (tests/golden/roipool.npz) and to the torch restatement of those
    paths in oracle/roipool.py (which is bit-exact against the same outputs on the CPU).
  * ``use_cuda=True``: the rotated RoIAlign kernel; axis-aligned RoIs go through it as theta = 0 boxes.  Two behaviours of the reference's
    op path are corrected by default and available verbatim with ``reference_op_quirks=True`` (or NRPN_ROIPOOL_REFERENCE_QUIRKS=1): it hands the box heading in RADIANS to an
    op that reads DEGREES and rotates sample points by -theta (ROIAlignRotated3D_cuda.cu:103,146-147; here: converted, so the sampled
    region IS the box), and it concatenates the pooled rows level by level (detector.py:250-259) while labels and RoIs stay in sample
    order (here: rows are returned in the order of the RoIs).
"""
import math
import os
from typing import List

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn, Tensor

from .. import ops
from . import hip_nn
from .coder import AABBCoder
from .coder.rotated_coder import RotatedCoder
from .level_mapper import _setup_scales
from .rotated_align import ROIAlignRotated3D


class ProposalTargetLayer(nn.Module):
    """Labels / sampled RoIs / matched ground truth per scene (reference detector.py:12-167).  RoI rows are
    (level index, box[6|7]); numpy's global RNG drives the sampling exactly as in the reference."""

    def __init__(self, nclasses, batch_size=1000, fg_fraction=0.5, fg_threshold=0.5, bg_threshold=0.2, is_rotated_bbox=False):
        super().__init__()
        self._num_classes = nclasses
        self.batch_size, self.fg_fraction = batch_size, fg_fraction
        self.fg_threshold, self.bg_threshold = fg_threshold, bg_threshold
        self.is_rotated_bbox = is_rotated_bbox
        self.bbox_size = 7 if is_rotated_bbox else 6

    def forward(self, all_rois, gt_boxes, gt_labels, is_sample):
        assert len(all_rois) == len(gt_boxes) == len(gt_labels)
        rois_per_image = int(self.batch_size / len(all_rois))
        fg_rois_per_image = max(1, int(np.round(self.fg_fraction * rois_per_image)))
        return self._sample_rois_pytorch(all_rois, gt_boxes, gt_labels, fg_rois_per_image, rois_per_image, self._num_classes, is_sample)

    def _sample_rois_pytorch(self, all_rois, gt_boxes, gt_label, fg_rois_per_image, rois_per_image, num_classes, is_sample):
        n = len(all_rois)
        max_overlaps, gt_assignment, labels = [], [], []
        for i in range(n):
            iou = ops.iou3d_matrix(all_rois[i][..., 1:].float(), gt_boxes[i].float())        # [R, G] in one launch
            mo, ga = torch.max(iou, 1)
            max_overlaps.append(mo)
            gt_assignment.append(ga)
            labels.append(gt_label[i].to(mo.device)[ga])
        if not is_sample:
            labels_batch, rois_batch, gt_rois_batch = [], [], []
            for i in range(n):
                lab = labels[0].new_zeros(all_rois[i].size(0))
                lab[max_overlaps[i] >= self.fg_threshold] = 1
                labels_batch.append(lab)
                rois_batch.append(all_rois[i])
                gt_rois_batch.append(gt_boxes[i][gt_assignment[i]])
            return labels_batch, rois_batch, gt_rois_batch
        labels_batch = labels[0].new_zeros(n, rois_per_image)
        rois_batch = all_rois[0].new_zeros(n, rois_per_image, all_rois[0].shape[-1])
        gt_rois_batch = all_rois[0].new_zeros(n, rois_per_image, gt_boxes[0].shape[-1])
        for i in range(n):
            fg_inds = torch.nonzero(max_overlaps[i] >= self.fg_threshold).view(-1)
            bg_inds = torch.nonzero(max_overlaps[i] < self.bg_threshold).view(-1)
            nfg, nbg = fg_inds.numel(), bg_inds.numel()
            dev = fg_inds.device

            def draw(count, upper):          # the reference's np.floor(np.random.rand(count) * upper) with-replacement draw
                return torch.from_numpy(np.floor(np.random.rand(count) * upper)).long().to(dev)
            if nfg > 0 and nbg > 0:
                fg_this = min(fg_rois_per_image, nfg)
                fg_inds = fg_inds[torch.from_numpy(np.random.permutation(nfg)).long().to(dev)[:fg_this]]
                bg_inds = bg_inds[draw(rois_per_image - fg_this, nbg)]
            elif nfg > 0:
                fg_inds, fg_this = fg_inds[draw(rois_per_image, nfg)], rois_per_image
                bg_inds = bg_inds[:0]
            elif nbg > 0:
                bg_inds, fg_this = bg_inds[draw(rois_per_image, nbg)], 0
                fg_inds = fg_inds[:0]
            else:
                raise ValueError("bg_num_rois = 0 and fg_num_rois = 0, this should not happen!")
            keep = torch.cat([fg_inds, bg_inds], 0)
            labels_batch[i].copy_(labels[i][keep])
            if fg_this < rois_per_image:
                labels_batch[i][fg_this:] = 0
            rois_batch[i] = all_rois[i][keep]
            gt_rois_batch[i] = gt_boxes[i][gt_assignment[i][keep]]
        return labels_batch, rois_batch, gt_rois_batch


class ROIPool(nn.Module):
    """Per-RoI pyramid features [R, C, *output_size] (reference detector.py:170-438).  ``spatial_scale`` = voxels of the input grid per
    feature voxel at each level; RoI rows are (level index, box)."""

    def __init__(self, output_size=(1, 1, 1), spatial_scale=(1, 1, 1, 1), enlarge_scale=0.2, is_rotated_bbox=False,
                 feature_extracting_type="pooling", max_res=200, remap=False, use_cuda=False, reference_op_quirks=None, aabb_use_kernel=False):
        super().__init__()
        self.output_size = [int(v) for v in output_size]
        self.spatial_scale = list(spatial_scale)
        self.enlarge_scale = enlarge_scale
        self.is_rotated_bbox = is_rotated_bbox
        self.feature_extracting_type = feature_extracting_type
        self.canonical_scale, self.canonical_level = max_res, len(spatial_scale)
        self.remap = remap
        if feature_extracting_type not in ("pooling", "interpolation"):
            raise NameError("Unkown feature_extracting_type")
        # use_cuda=True: the rotated RoIAlign op (the reference's --use_cuda path, :247-261).  use_cuda=False (the reference CLI's DEFAULT):
        # its torch paths -- integer crops + adaptive max-pool for AABBs (:397-438), a rotated 8-corner gather followed by max-pool or a
        # trilinear resize for OBBs (:264-395) -- as HIP kernels (csrc/roipool.hip), so that an RCNN trained with the reference's default
        # pooling reproduces its scores.
        # Dispatch as the reference (:239-245): the op serves ROTATED RoIs when use_cuda is set; axis-aligned RoIs always take the integer
        # crop + adaptive max-pool (normal_forward), whatever use_cuda says -- AABB RCNN weights trained with the reference expect that
        # pooling.  ``aabb_use_kernel=True`` is an explicit opt-in (not in the reference) that sends AABBs through the RoIAlign kernel as
        # theta = 0 boxes with a true (1 + e) enlargement.  The default of use_cuda is the reference's (False).
        self.use_cuda = bool(use_cuda)
        self.aabb_use_kernel = bool(aabb_use_kernel)
        if reference_op_quirks is None:
            reference_op_quirks = os.environ.get("NRPN_ROIPOOL_REFERENCE_QUIRKS", "0") == "1"
        self.reference_op_quirks = bool(reference_op_quirks)
        self.align = ROIAlignRotated3D(self.output_size, sampling_ratio=0)

    def enlarge_roi(self, roi):
        """Boxes -> (x, y, z, w, l, h, theta) with the extent grown by ``enlarge_scale`` (reference :195-211)."""
        if self.is_rotated_bbox:
            out = roi.clone()
            out[..., 3:6] = out[..., 3:6] * (1 + self.enlarge_scale)
            return out
        ctr = (roi[..., 3:] + roi[..., :3]) / 2
        ext = (roi[..., 3:] - roi[..., :3]) * (1 + self.enlarge_scale)
        return torch.cat([ctr, ext, torch.zeros_like(ctr[..., :1])], dim=-1)

    def forward(self, feature, rois, original_size=None):
        if self.remap:
            scales = [1 / s for s in self.spatial_scale]
            mapper = _setup_scales(scales, self.canonical_scale, self.canonical_level)
            remapped = []
            for r in rois:
                shape = r.shape
                flat = r.reshape(-1, shape[-1])
                boxes = flat[..., 1:]
                geo = boxes if self.is_rotated_bbox else torch.cat([(boxes[..., 3:] + boxes[..., :3]) / 2, boxes[..., 3:] - boxes[..., :3]], -1)
                lv = mapper(geo).to(flat.dtype)
                remapped.append(torch.cat([lv[..., None], boxes], -1).reshape(shape))
            rois = remapped
        if not self.is_rotated_bbox and not (self.use_cuda and self.aabb_use_kernel):
            return self._normal_forward(feature, rois)
        if not self.use_cuda:
            return self._rotated_forward(feature, rois)
        out = []
        for f, r in zip(feature, rois):
            r = r.reshape(-1, r.shape[-1])
            lv = r[..., 0].long()
            obb = self.enlarge_roi(r[..., 1:]).float()
            pooled = None
            for l in range(self.canonical_level):
                sel = torch.nonzero(lv == l).view(-1)
                if sel.numel() == 0:
                    continue
                rows = obb[sel]
                # op rows: (batch index, centre, extent, angle in degrees); its sampling rotates by -angle, a box with heading theta
                # is therefore sampled with angle = -theta (see the module docstring)
                angle = rows[:, 6:7] if self.reference_op_quirks else -rows[:, 6:7] * (180.0 / math.pi)
                op_rois = torch.cat([torch.zeros_like(rows[:, :1]), rows[:, :6], angle], dim=1)
                feat = self.align(f[l][None], op_rois, float(1 / self.spatial_scale[l]))
                if self.reference_op_quirks:       # level-major rows, as the reference's torch.cat over levels leaves them
                    pooled = feat if pooled is None else torch.cat([pooled, feat])
                    continue
                if pooled is None:
                    pooled = feat.new_zeros((r.shape[0],) + tuple(feat.shape[1:]))
                pooled = pooled.index_copy(0, sel, feat)
            if pooled is None:
                c = f[0].shape[0]
                pooled = f[0].new_zeros((0, c, *self.output_size))
            out.append(pooled)
        return out


    # ---------------------------------------------------------------------------------------------- the reference's default pooling (use_cuda=False)
    def _level_dims(self, f, device):
        return torch.tensor([[int(v) for v in lv.shape[1:]] for lv in f], dtype=torch.long, device=device)

    def _normal_forward(self, features, rois):
        """AABB RoIs (reference :397-438): the box placed at centre -+ 0.5 * (half extent * (1 + enlarge)) -- the reference's "enlargement" pools
        (1 + e) / 2 of the RoI; kept, it is what its RCNN weights were trained on -- in level voxels, floor of both corners, the inclusive
        integer crop with python slicing semantics (the high end is clipped to the map), adaptive max-pool.  Crops are computed with a few
        tensor ops on the device (no host read-back); cropping + pooling of ALL RoIs of a scene = one HIP launch per level (csrc/roipool.hip)."""
        out = []
        for f, r in zip(features, rois):
            r = r.reshape(-1, r.shape[-1])
            if r.shape[0] == 0:
                out.append(f[0].new_zeros((0, f[0].shape[0], *self.output_size), dtype=torch.float32))
                continue
            lv = r[..., 0].long()
            scale = torch.tensor(self.spatial_scale, dtype=r.dtype, device=r.device)[lv][:, None]
            roi = r[..., 1:]
            ext = (roi[..., 3:] - roi[..., :3]) / 2 * (1 + self.enlarge_scale)
            ctr = (roi[..., 3:] + roi[..., :3]) / 2
            pos = torch.floor(torch.cat([ctr - 0.5 * ext, ctr + 0.5 * ext], dim=-1) / scale).long()
            dims = self._level_dims(f, r.device)[lv]
            lo, hi = pos[:, :3], pos[:, 3:] + 1                      # python slice lo : hi
            start = torch.where(lo < 0, (dims + lo).clamp_min(0), torch.minimum(lo, dims))
            stop = torch.where(hi < 0, (dims + hi).clamp_min(0), torch.minimum(hi, dims))
            crop = torch.cat([start, (stop - start).clamp_min(0)], dim=1).int()
            out.append(ops.RoiPoolFn.apply("aabb", crop, lv.int(), self.spatial_scale, self.output_size, *f))
        return out

    def _rotated_forward(self, features, rois):
        """OBB RoIs without the op (reference :264-395): a regular grid of ceil(extent / scale) points per RoI, rotated by theta about the box
        centre, each point the reference's 8-corner blend (not a trilinear interpolation; kept as it is), zero outside the map; then adaptive
        max-pool ('pooling') or a trilinear resize ('interpolation') -- one HIP launch per level over all RoIs of a scene (csrc/roipool.hip).
        Like the reference, the RoI extents are enlarged IN PLACE (its ``enlarge_roi`` writes through the view it is given), so the caller's
        RoIs -- and the proposals decoded from them afterwards -- are the enlarged ones."""
        out = []
        for f, r in zip(features, rois):
            flat = r.reshape(-1, r.shape[-1])              # a view: the in-place enlargement below reaches the caller's tensor
            flat[..., 4:7] = flat[..., 4:7] * (1 + self.enlarge_scale)
            if flat.shape[0] == 0:
                out.append(f[0].new_zeros((0, f[0].shape[0], *self.output_size), dtype=torch.float32))
                continue
            out.append(ops.RoiPoolFn.apply(self.feature_extracting_type, flat[..., 1:].detach().float().contiguous(), flat[..., 0].int(),
                                           self.spatial_scale, self.output_size, *f))
        return out


class RCNN(nn.Module):
    """Two optional 3x3x3 conv+ReLU layers on the pooled features, then Linear classification / box heads (reference :441-496)."""

    def __init__(self, input_dim, block, n_classes, input_size, is_add_layer=False, is_rotated_bbox=False, is_flatten=True):
        super().__init__()
        self.in_planes, self.n_classes = input_dim, n_classes
        self.is_rotated_bbox, self.is_flatten = is_rotated_bbox, is_flatten
        self.reg_dim = 7 if is_rotated_bbox else 6
        self.layer = None
        if is_add_layer:
            convs = []
            for _ in range(2):
                convs += [nn.Conv3d(input_dim, input_dim, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            self.layer = nn.Sequential(*convs)
        width = input_dim * (input_size[0] * input_size[1] * input_size[2] if is_flatten else 1)
        self.flatten_size = input_size[0] * input_size[1] * input_size[2]
        self.RCNN_bbox_pred = nn.Linear(width, self.reg_dim)
        self.RCNN_cls_score = nn.Linear(width, self.n_classes)
        for m in (self.RCNN_bbox_pred, self.RCNN_cls_score):      # rows of ONE fused GEMM: a flat-arena trainer keeps them in the reference layout
            m.__dict__["_nrpn_fused_gemm"] = True
        self._pack = ops.PackedWeight()

    def forward(self, pooling_feature):
        x = pooling_feature
        if self.layer is not None and x.shape[0] > 0:
            # every pooled RoI is one tiny grid: the batch dimension of the conv kernels (channels-last, zero padding per grid)
            cl = hip_nn.run_modules(self.layer, hip_nn.as_ndhwc(x, x.dtype))
            x = hip_nn.as_ncdhw(cl)
        x = x.reshape(x.size(0), -1) if self.is_flatten else x.mean(-1).mean(-1).mean(-1)
        x = x.float()
        if x.shape[0] == 0 or (x.shape[1] * 4) % 64:
            return F.linear(x, self.RCNN_bbox_pred.weight, self.RCNN_bbox_pred.bias), F.linear(x, self.RCNN_cls_score.weight, self.RCNN_cls_score.bias)
        # both Linear heads as ONE one-tap GEMM on the MFMA kernels (rows = [bbox_pred | cls_score | zero padding to 64]), fp32 (exact
        # fp32 MFMA chains): the same construction as the RPN head's fused cls / bbox GEMM
        rows = ((self.reg_dim + self.n_classes + 63) // 64) * 64
        y = ops.ConvFn.apply(x.reshape(x.shape[0], 1, 1, 1, x.shape[1]).contiguous(), self._pack, rows, False, True, 2,
                             self.RCNN_bbox_pred.weight, self.RCNN_cls_score.weight, self.RCNN_bbox_pred.bias, self.RCNN_cls_score.bias)
        y = y.reshape(x.shape[0], rows)
        return y[:, :self.reg_dim], y[:, self.reg_dim:self.reg_dim + self.n_classes]


class Classification_Model(nn.Module):
    def __init__(self, feature_extractor, sample_model, pooling_model, RCNN_model, n_classes=2, is_training=True, batch_size=20,
                 is_rotated_bbox=False):
        super().__init__()
        self.batch_size = batch_size
        self.feature_extractor, self.sample_model = feature_extractor, sample_model
        self.pooling_model, self.RCNN_model = pooling_model, RCNN_model
        self.num_class, self.score_thresh = n_classes, 0.7
        self.is_training, self.is_rotated_bbox = is_training, is_rotated_bbox
        self.reg_dim = 7 if is_rotated_bbox else 6
        self.box_coder = RotatedCoder() if is_rotated_bbox else AABBCoder()

    def transform(self, meshes):
        if len(meshes) > 1:
            target = np.max([m.shape for m in meshes], axis=0)
            meshes = [F.pad(m, (0, int(target[-1] - m.shape[-1]), 0, int(target[-2] - m.shape[-2]), 0, int(target[-3] - m.shape[-3])))
                      for m in meshes]
        return meshes

    def compute_loss(self, pred_scores, pred_bbox_deltas, gt_labels, regression_targets):
        gt_labels = gt_labels.reshape(-1).long()
        obj = F.cross_entropy(pred_scores, gt_labels)
        inds = torch.nonzero(gt_labels > 0).view(-1)
        reg_t = regression_targets.view(-1, self.reg_dim)
        if inds.numel():
            box = F.smooth_l1_loss(pred_bbox_deltas[inds], reg_t[inds], beta=1 / 9, reduction="sum") / inds.numel()
        else:
            box = torch.zeros((), dtype=reg_t.dtype, device=pred_scores.device)
        return {"loss_objectness": obj, "loss_rpn_box_reg": box}

    def forward(self, rois, gt_bboxes, gt_bbox_labels, features, is_sample=True, is_reg=False):
        original_size = []
        if self.feature_extractor is not None:            # fine-tuning: ``features`` are the raw rgb-sigma grids
            flat = self.transform([f[0] for f in features])
            original_size = [f.shape[1:] for f in flat]
            maps = list(self.feature_extractor(ops.stack_scenes(flat)))
            features = [[m[i] for m in maps] for i in range(len(flat))]
        gt_labels, sample_rois, gt_bbox = self.sample_model(rois, gt_bboxes, gt_bbox_labels, is_sample=is_sample)
        pooled = self.pooling_model(features, sample_rois, original_size)
        counts = [p.size(0) for p in pooled]
        pred_deltas, pred_scores = self.RCNN_model(torch.cat(pooled, dim=0))
        cls_prob = F.softmax(pred_scores, 1)
        n = gt_bbox.size(0) if is_sample else len(gt_bbox)
        deltas_b, prob_b = list(pred_deltas.split(counts)), list(cls_prob.split(counts))
        rois_b = [sample_rois[i][..., 1:] for i in range(n)]
        proposals = [self.box_coder.decode_single(deltas_b[i], rois_b[i].reshape(-1, rois_b[i].shape[-1]).float()) for i in range(n)]
        loss = 0.
        if is_sample:
            reg_t = self.box_coder.encode_single(gt_bbox.reshape(-1, gt_bbox.size(-1)).float(),
                                                 sample_rois[..., 1:].reshape(-1, sample_rois.size(-1) - 1).float())
            loss = self.compute_loss(pred_scores, pred_deltas, gt_labels, reg_t.reshape(n, sample_rois.size(1), -1))
        if is_reg:
            return [proposals, gt_labels], prob_b, loss
        return [rois_b, gt_labels], prob_b, loss

"""FCOS post-processing on the HIP kernels (reference nerf_rpn/model/fcos/inference.py:11-195): per-level candidate
selection, decode, all-level single-class NMS, kth-value cap."""
import torch

from ... import ops


class FCOSPostProcessor(torch.nn.Module):
    def __init__(self, pre_nms_thresh, pre_nms_top_n, nms_thresh, fpn_post_nms_top_n, min_size, num_classes, bbox_aug_enabled=False,
                 use_obb=False):
        super().__init__()
        self.pre_nms_thresh, self.pre_nms_top_n, self.nms_thresh = pre_nms_thresh, pre_nms_top_n, nms_thresh
        self.fpn_post_nms_top_n, self.min_size, self.num_classes = fpn_post_nms_top_n, min_size, num_classes
        self.bbox_aug_enabled, self.use_obb = bbox_aug_enabled, use_obb

    def forward(self, geom, logits, reg, ctr, grid_sizes, pad_sizes):
        """logits / ctr [total], reg [total, 6|8] in the flattened (level, scene, voxel) order -> per scene
        (boxes [K, 1 + 6|7] with the level index in column 0, scores [K]) in score-descending order."""
        if self.bbox_aug_enabled:
            raise NotImplementedError("bbox_aug_enabled is never set by run_fcos.py")
        n, L = geom.n, geom.levels
        D = 8 if self.use_obb else 6
        k = min(int(self.pre_nms_top_n), max(geom.counts))
        if k * L > 16384:
            # all levels of a scene are sorted and suppressed together in one workgroup's LDS (<= 16384 rows); the reference has no such
            # limit, so refuse loudly instead of silently clamping (as RegionProposalNetwork.filter_proposals does)
            raise ValueError(f"pre_nms_top_n ({self.pre_nms_top_n}) x pyramid levels ({L}) must stay <= 16384 on the HIP path")
        scores = ops.fcos_scores(geom, logits, ctr, pad_sizes, self.pre_nms_thresh)
        idx, val = ops.segmented_topk(scores, geom.segment_offsets, k)          # [(level, scene), k], score-descending
        seg_start = torch.tensor(geom.segment_offsets[:-1], dtype=torch.int32, device=idx.device)
        local = torch.where(idx >= 0, idx - seg_start[:, None], idx).contiguous()
        boxes, sc, lv = ops.fcos_decode(geom, local, val.contiguous(), reg, grid_sizes, D, self.min_size)
        boxes, sc, lv = boxes.view(L, n, k, -1), sc.view(L, n, k), lv.view(L, n, k)
        pending = []
        cap = min(L * k, 16384)
        zeros = torch.zeros(L * k, dtype=torch.int32, device=idx.device)
        for i in range(n):
            b, s, l = boxes[:, i].reshape(L * k, -1), sc[:, i].reshape(-1).contiguous(), lv[:, i].reshape(-1)
            order = ops.argsort_desc(s)                                          # dropped slots carry -1 and sort last
            b, s, l = b[order].contiguous(), s[order].contiguous(), l[order].to(torch.int32).contiguous()
            cnt = (s >= 0).sum().to(torch.int32).reshape(1)
            keep = ops.nms3d_sorted(b, zeros, self.nms_thresh, cnt)              # single class over all levels
            pending.append(ops.select_kept(b, s, l, keep, cnt, cap))
        out_boxes, out_scores = [], []
        for ob, os_, ol, oc in pending:
            m = int(oc.item())
            top = self.fpn_post_nms_top_n
            if m > top > 0:       # kthvalue cap keeps every score >= the top-th best (ties included), inference.py:176-186
                m = int((os_[:m] >= os_[top - 1]).sum().item())
            out_boxes.append(torch.cat([ol[:m, None], ob[:m]], dim=-1))
            out_scores.append(os_[:m])
        return out_boxes, out_scores

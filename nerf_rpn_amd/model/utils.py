"""Box utilities with the reference's names (nerf_rpn/model/utils.py), backed by the HIP kernels."""
from typing import List, Tuple

import torch
from torch import Tensor

from .. import ops


def box_iou_3d(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """[N,6|7] x [M,6|7] -> [N,M] (utils.py:387-415)."""
    return ops.iou3d_matrix(boxes1, boxes2)


@torch.no_grad()
def batched_box_iou(boxes1: Tensor, boxes2: Tensor, batch_size=16) -> Tensor:
    return ops.iou3d_matrix(boxes1, boxes2)   # chunking was an OOM workaround; one kernel needs none


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Greedy NMS; returns kept indices in score-descending order (utils.py:215-230)."""
    if boxes.shape[0] == 0:
        return torch.empty(0, dtype=torch.long, device=boxes.device)
    order = ops.argsort_desc(scores)
    keep = ops.nms3d_sorted(boxes[order], None, iou_threshold)
    return order[keep.bool()]


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float) -> Tensor:
    """Per-category NMS; kept indices sorted by decreasing score (utils.py:233-265)."""
    if boxes.shape[0] == 0:
        return torch.empty(0, dtype=torch.long, device=boxes.device)
    by_score = ops.argsort_desc(scores)
    lv, rel = torch.sort(idxs[by_score], stable=True)
    order = by_score[rel]
    remap = torch.unique_consecutive(lv, return_inverse=True)[1].to(torch.int32)
    keep = ops.nms3d_sorted(boxes[order], remap.contiguous(), iou_threshold)
    kept = order[keep.bool()]
    return kept[ops.argsort_desc(scores[kept])]


def remove_small_boxes(boxes: Tensor, min_size: float) -> Tensor:
    e = boxes[:, 3:6] - boxes[:, 0:3] if boxes.size(1) == 6 else boxes[:, 3:6]
    return torch.where((e >= min_size).all(dim=1))[0]


def clip_boxes_to_mesh(boxes: Tensor, size: Tuple[int, int, int]) -> Tensor:
    if boxes.size(1) == 6:
        hi = torch.tensor(list(size) * 2, dtype=boxes.dtype, device=boxes.device)
        return torch.min(boxes.clamp(min=0), hi)
    c = boxes[:, :3]
    hi = torch.tensor(list(size), dtype=boxes.dtype, device=boxes.device)
    return boxes[((c >= 0) & (c <= hi)).all(dim=1)]


class Matcher:
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold: float, low_threshold: float, allow_low_quality_matches: bool = False) -> None:
        torch._assert(low_threshold <= high_threshold, "low_threshold should be <= high_threshold")
        self.high_threshold, self.low_threshold = high_threshold, low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix: Tensor) -> Tensor:
        """Generic matrix form (index logic only); the RPN hot path uses the fused ``ops.match_anchors`` instead."""
        if match_quality_matrix.numel() == 0:
            raise ValueError("No ground-truth or proposal boxes available for one of the images during training")
        vals, matches = match_quality_matrix.max(dim=0)
        best = matches.clone()
        matches[vals < self.low_threshold] = self.BELOW_LOW_THRESHOLD
        matches[(vals >= self.low_threshold) & (vals < self.high_threshold)] = self.BETWEEN_THRESHOLDS
        if self.allow_low_quality_matches:
            top = match_quality_matrix.max(dim=1)[0]
            cols = torch.where(match_quality_matrix == top[:, None])[1]
            matches[cols] = best[cols]
        return matches


class BalancedPositiveNegativeSampler:
    def __init__(self, batch_size_per_image: int, positive_fraction: float) -> None:
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction

    def sample_indices(self, labels: Tensor):
        """-> (pos_idx, neg_idx) int64, each sorted ascending (what torch.where on the reference's masks yields)."""
        pos = torch.where(labels >= 1)[0]
        neg = torch.where(labels == 0)[0]
        n_pos = min(pos.numel(), int(self.batch_size_per_image * self.positive_fraction))
        n_neg = min(neg.numel(), self.batch_size_per_image - n_pos)
        p = pos[torch.randperm(pos.numel(), device=pos.device)[:n_pos]]
        q = neg[torch.randperm(neg.numel(), device=neg.device)[:n_neg]]
        return p.sort()[0], q.sort()[0]

    def sample_batch(self, labels: List[Tensor], extra_flags=None, before_readback=None):
        """All scenes of a batch -> ([(pos_idx, neg_idx)], host values of ``extra_flags``).  On the device this is the fused sampler
        kernel (ops.sample_pos_neg: no torch.where / randperm, ONE host read-back for the whole batch, which also carries the caller's
        pending 0-dim flags); the draw is seeded from torch's CPU generator, so torch.manual_seed makes it reproducible."""
        if labels and labels[0].is_cuda:
            from .. import ops
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            return ops.sample_pos_neg([lab.float() for lab in labels], self.batch_size_per_image,
                                      int(self.batch_size_per_image * self.positive_fraction), seed, extra_flags, before_readback)
        return [self.sample_indices(lab) for lab in labels], [bool(f) for f in (extra_flags or [])]

    def __call__(self, matched_idxs: List[Tensor]):
        pos_masks, neg_masks = [], []
        for lab in matched_idxs:
            p, q = self.sample_indices(lab)
            pm = torch.zeros_like(lab, dtype=torch.uint8)
            nm = torch.zeros_like(lab, dtype=torch.uint8)
            pm[p] = 1
            nm[q] = 1
            pos_masks.append(pm)
            neg_masks.append(nm)
        return pos_masks, neg_masks

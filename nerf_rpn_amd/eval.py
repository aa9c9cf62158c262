"""Proposal metrics with the reference's function names and return dicts (reference nerf_rpn/eval.py:14-81, 319-395), on the device:

  recall : one fused IoU-matrix launch per scene (``ops.iou3d_matrix``) + one launch of the greedy GT <-> proposal matching
           (``nrpn_recall_match_f32``: per-GT maxima in LDS, min(P, G) rounds) -- the reference runs min(P, G) rounds of two full
           matrix reductions in Python per scene;
  AP     : IoU matrices and per-detection best GT per scene, ONE global sort of all detections, and the "first detection that claims a
           ground-truth box is the true positive" rule as two integer-atomic launches (``nrpn_ap_mark``) instead of a Python loop over every
           detection with one rotated-IoU call each; cumulative sums and the precision envelope are device scans.
Inputs may live on the host or the device; results come back as host tensors, as the reference returns them."""
import torch

from . import ops
from .lib import call


def _dev(*ts):
    for t in ts:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return torch.device("cuda", torch.cuda.current_device())


def _iou(proposals, gt, dev):
    return ops.iou3d_matrix(proposals.to(dev).float(), gt.to(dev).float())


def recall_match(overlaps):
    """overlaps [P, G] (device, consumed) -> covered [min(P, G)]: the overlaps recorded by the reference's greedy matching (eval.py:41-61)."""
    p, g = overlaps.shape
    covered = torch.zeros(g, dtype=torch.float32, device=overlaps.device)
    if p and g:
        ov = overlaps.contiguous()
        call("recall_match_f32", ov.data_ptr(), p, g, covered.data_ptr(), ops._s())
    return covered


def evaluate_box_proposals_recall(proposals_list, proposal_scores_list, gt_boxes_list, thresholds=None, limit=None):
    gt_overlaps = []
    num_pos = 0
    for proposals, scores, gt_boxes in zip(proposals_list, proposal_scores_list, gt_boxes_list):
        if proposals.shape[0] == 0 or gt_boxes.shape[0] == 0:
            continue
        dev = _dev(proposals, gt_boxes)
        order = torch.argsort(scores.to(dev), descending=True, stable=True)
        proposals = proposals.to(dev)[order]
        num_pos += gt_boxes.shape[0]
        if limit is not None and len(proposals) > limit:
            proposals = proposals[:limit]
        gt_overlaps.append(recall_match(_iou(proposals, gt_boxes, dev)))
    gt_overlaps = torch.cat(gt_overlaps, dim=0).cpu() if gt_overlaps else torch.zeros(0, dtype=torch.float32)
    gt_overlaps, _ = torch.sort(gt_overlaps)
    if thresholds is None:
        thresholds = torch.arange(0.5, 0.95 + 1e-5, 0.05, dtype=torch.float32)
    recalls = torch.zeros_like(thresholds)
    for i, t in enumerate(thresholds):
        recalls[i] = (gt_overlaps >= t).float().sum() / float(num_pos)
    return {"ar": recalls.mean(), "recalls": recalls, "thresholds": thresholds, "gt_overlaps": gt_overlaps, "num_pos": num_pos}


def evaluate_box_proposals_ap(proposals_list, proposal_scores_list, gt_boxes_list, iou_thresh=0.25, top_k=None):
    """Pascal-VOC AP at one IoU threshold (eval.py:319-395)."""
    num_gt = 0
    dev = _dev(*proposals_list, *gt_boxes_list)
    gmax = max([int(g.shape[0]) for g in gt_boxes_list] + [1])
    scores_all, best_iou, keys = [], [], []
    for i, (proposals, scores, gt_boxes) in enumerate(zip(proposals_list, proposal_scores_list, gt_boxes_list)):
        proposals, scores = proposals.to(dev), scores.to(dev)
        if top_k is not None and len(proposals) > top_k:
            ids = torch.argsort(scores, descending=True, stable=True)[:top_k]
            proposals, scores = proposals[ids], scores[ids]
        num_gt += gt_boxes.shape[0]
        if len(proposals) == 0:
            continue
        if gt_boxes.shape[0] == 0:      # every detection of a scene without ground truth is a false positive
            m = torch.full((len(proposals),), -1.0, device=dev)
            a = torch.zeros(len(proposals), dtype=torch.int64, device=dev)
        else:
            m, a = _iou(proposals, gt_boxes, dev).max(dim=1)        # one launch per scene instead of one per detection
        scores_all.append(scores.float())
        best_iou.append(m)
        keys.append(a + i * gmax)
    scores_all, best_iou, keys = torch.cat(scores_all), torch.cat(best_iou).contiguous(), torch.cat(keys).contiguous()
    n = scores_all.numel()
    order = torch.argsort(scores_all, descending=True, stable=True).contiguous()
    first = torch.empty(len(gt_boxes_list) * gmax, dtype=torch.int32, device=dev)
    tpm = torch.empty(n, dtype=torch.uint8, device=dev)
    call("ap_mark", order.data_ptr(), best_iou.data_ptr(), keys.data_ptr(), n, first.numel(), float(iou_thresh), first.data_ptr(), tpm.data_ptr(),
         ops._s())
    tp = torch.cumsum(tpm.long(), dim=0)
    fp = torch.cumsum(1 - tpm.long(), dim=0)
    recalls = tp / num_gt
    precisions = tp / (tp + fp)
    zero, one = recalls.new_zeros(1), recalls.new_ones(1)
    mrec = torch.cat((zero, recalls, one))
    mpre = torch.cat((zero, precisions, zero))
    mpre = torch.flip(torch.cummax(torch.flip(mpre, (0,)), dim=0).values, (0,))       # precision envelope (eval.py:386-388)
    idx = torch.where(mrec[1:] != mrec[:-1])[0]
    ap = torch.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1])
    return {"ap": ap.cpu(), "precisions": precisions.cpu(), "recalls": recalls.cpu(), "thresholds": iou_thresh, "num_det": (tp + fp).cpu()}

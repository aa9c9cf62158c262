"""Proposal metrics with the reference's function names and return dicts (reference nerf_rpn/eval.py:14-81, 319-395).

The IoU matrices (the expensive part: rotated IoU of up to 2500 proposals x G boxes per scene) come from one fused HIP kernel
launch per scene (``ops.iou3d_matrix``); the greedy matching that follows is index bookkeeping on the small host copy."""
import torch

from . import ops


def _iou(proposals, gt):
    dev = proposals.device if proposals.is_cuda else (gt.device if gt.is_cuda else torch.device("cuda"))
    return ops.iou3d_matrix(proposals.to(dev).float(), gt.to(dev).float()).cpu()


def evaluate_box_proposals_recall(proposals_list, proposal_scores_list, gt_boxes_list, thresholds=None, limit=None):
    gt_overlaps = []
    num_pos = 0
    for proposals, scores, gt_boxes in zip(proposals_list, proposal_scores_list, gt_boxes_list):
        order = torch.argsort(scores, descending=True)
        proposals = proposals[order]
        if proposals.shape[0] == 0 or gt_boxes.shape[0] == 0:
            continue
        num_pos += gt_boxes.shape[0]
        if limit is not None and len(proposals) > limit:
            proposals = proposals[:limit]
        overlaps = _iou(proposals, gt_boxes)
        covered = torch.zeros(gt_boxes.shape[0])
        for j in range(min(proposals.shape[0], gt_boxes.shape[0])):
            best_per_gt, arg_per_gt = overlaps.max(dim=0)
            gt_ovr, gt_ind = best_per_gt.max(dim=0)
            assert gt_ovr >= 0
            box_ind = arg_per_gt[gt_ind]
            covered[j] = overlaps[box_ind, gt_ind]
            overlaps[box_ind, :] = -1
            overlaps[:, gt_ind] = -1
        gt_overlaps.append(covered)
    gt_overlaps = torch.cat(gt_overlaps, dim=0) if gt_overlaps else torch.zeros(0, dtype=torch.float32)
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
    scene_ids, scores_all, best_iou, best_gt = [], [], [], []
    for i, (proposals, scores, gt_boxes) in enumerate(zip(proposals_list, proposal_scores_list, gt_boxes_list)):
        if top_k is not None and len(proposals) > top_k:
            ids = torch.argsort(scores, descending=True)[:top_k]
            proposals, scores = proposals[ids], scores[ids]
        num_gt += gt_boxes.shape[0]
        if len(proposals) == 0:
            continue
        ov = _iou(proposals, gt_boxes)                      # one launch per scene instead of one per detection
        m, a = ov.max(dim=1)
        scene_ids.append(torch.full((len(proposals),), i, dtype=torch.int64))
        scores_all.append(scores.cpu())
        best_iou.append(m)
        best_gt.append(a)
    scene_ids, scores_all = torch.cat(scene_ids), torch.cat(scores_all)
    best_iou, best_gt = torch.cat(best_iou), torch.cat(best_gt)
    order = torch.argsort(scores_all, descending=True)
    scene_ids, best_iou, best_gt = scene_ids[order], best_iou[order], best_gt[order]
    used = [torch.zeros(len(g), dtype=torch.bool) for g in gt_boxes_list]
    tp = torch.zeros(len(order), dtype=torch.bool)
    fp = torch.zeros(len(order), dtype=torch.bool)
    for i in range(len(order)):
        s, g = int(scene_ids[i]), int(best_gt[i])
        if best_iou[i] > iou_thresh and not used[s][g]:
            tp[i] = True
            used[s][g] = True
        else:
            fp[i] = True
    tp, fp = torch.cumsum(tp, dim=0), torch.cumsum(fp, dim=0)
    recalls = tp / num_gt
    precisions = tp / (tp + fp)
    mrec = torch.cat((torch.tensor([0.0]), recalls, torch.tensor([1.0])))
    mpre = torch.cat((torch.tensor([0.0]), precisions, torch.tensor([0.0])))
    for i in range(mpre.size(0) - 1, 0, -1):
        mpre[i - 1] = torch.max(mpre[i - 1], mpre[i])
    idx = torch.where(mrec[1:] != mrec[:-1])[0]
    ap = torch.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1])
    return {"ap": ap, "precisions": precisions, "recalls": recalls, "thresholds": iou_thresh, "num_det": tp + fp}

"""ORACLE (test infrastructure only): box utilities of the RPN path on CPU.

Restates reference nerf_rpn/model/utils.py: sampler :35-95, Matcher :98-211, nms :215-230,
batched_nms :233-265, remove_small_boxes :268-289, clip_boxes_to_mesh :329-367,
batched_box_iou :370-384, box_iou_3d :387-415, AABB IoU :418-458.
"""
import torch

from . import geometry


def aabb_iou_matrix(a, b):
    """[N,6] x [M,6] -> [N,M]; utils.py:445-458."""
    va = (a[:, 3] - a[:, 0]) * (a[:, 4] - a[:, 1]) * (a[:, 5] - a[:, 2])
    vb = (b[:, 3] - b[:, 0]) * (b[:, 4] - b[:, 1]) * (b[:, 5] - b[:, 2])
    lo = torch.max(a[:, None, :3], b[:, :3])
    hi = torch.min(a[:, None, 3:], b[:, 3:])
    e = (hi - lo).clamp(min=0)
    inter = e[..., 0] * e[..., 1] * e[..., 2]
    return inter / (va[:, None] + vb - inter)


def iou_matrix(a, b):
    """utils.py:387-415 (AABB if 6 columns, OBB if 7)."""
    if a.shape[1] == 6 and b.shape[1] == 6:
        return aabb_iou_matrix(a, b)
    if a.shape[1] == 7 and b.shape[1] == 7:
        ar = a.unsqueeze(1).repeat(1, b.shape[0], 1)
        br = b.unsqueeze(0).repeat(a.shape[0], 1, 1)
        return geometry.iou_3d(ar, br).float()
    raise ValueError("box widths must both be 6 or both be 7")


def iou_matrix_chunked(a, b, chunk=16):
    """utils.py:370-384."""
    return torch.cat([iou_matrix(a[i:i + chunk], b) for i in range(0, a.shape[0], chunk)], dim=0)


def greedy_nms(boxes, scores, thr):
    """utils.py:215-230: keep order = score-descending; suppress IoU > thr."""
    order = scores.argsort(descending=True)
    keep = []
    while order.numel() > 0:
        i = order[0]
        keep.append(int(i))
        if order.numel() == 1:
            break
        iou = iou_matrix(boxes[i].unsqueeze(0), boxes[order[1:]]).reshape(-1)
        order = order[1:][iou <= thr]
    return torch.tensor(keep, dtype=torch.long)


def nms_per_level(boxes, scores, levels, thr):
    """utils.py:233-265."""
    mask = torch.zeros_like(scores, dtype=torch.bool)
    for lv in torch.unique(levels):
        idx = torch.where(levels == lv)[0]
        mask[idx[greedy_nms(boxes[idx], scores[idx], thr)]] = True
    kept = torch.where(mask)[0]
    return kept[scores[kept].sort(descending=True)[1]]


def big_enough(boxes, min_size):
    """utils.py:268-289 -> indices."""
    if boxes.shape[1] == 6:
        e = boxes[:, 3:6] - boxes[:, 0:3]
    else:
        e = boxes[:, 3:6]
    return torch.where((e >= min_size).all(dim=1))[0]


def clip_to_grid(boxes, size):
    """utils.py:329-367.  AABB: clamp; OBB: *drop rows* whose centre is outside (quirk B3)."""
    if boxes.shape[1] == 6:
        out = boxes.clone()
        for ax in range(3):
            out[:, ax] = boxes[:, ax].clamp(min=0, max=size[ax])
            out[:, ax + 3] = boxes[:, ax + 3].clamp(min=0, max=size[ax])
        return out
    ok = torch.ones(boxes.shape[0], dtype=torch.bool)
    for ax in range(3):
        ok &= (boxes[:, ax] >= 0) & (boxes[:, ax] <= size[ax])
    return boxes[ok]


BELOW, BETWEEN = -1, -2


def match(quality, hi, lo, allow_low_quality=True):
    """Matcher.__call__, utils.py:142-211.  quality: [G, A] -> int64 [A]."""
    if quality.numel() == 0:
        raise ValueError("empty match-quality matrix")
    vals, idx = quality.max(dim=0)
    best = idx.clone()
    idx[vals < lo] = BELOW
    idx[(vals >= lo) & (vals < hi)] = BETWEEN
    if allow_low_quality:
        top = quality.max(dim=1)[0]
        cols = torch.where(quality == top[:, None])[1]
        idx[cols] = best[cols]
    return idx


def sample_pos_neg(labels_list, per_scene, pos_fraction):
    """BalancedPositiveNegativeSampler, utils.py:35-95 (consumes torch's global RNG)."""
    pos_masks, neg_masks = [], []
    for lab in labels_list:
        pos = torch.where(lab >= 1)[0]
        neg = torch.where(lab == 0)[0]
        n_pos = min(pos.numel(), int(per_scene * pos_fraction))
        n_neg = min(neg.numel(), per_scene - n_pos)
        p1 = torch.randperm(pos.numel())[:n_pos]
        p2 = torch.randperm(neg.numel())[:n_neg]
        pm = torch.zeros_like(lab, dtype=torch.uint8)
        nm = torch.zeros_like(lab, dtype=torch.uint8)
        pm[pos[p1]] = 1
        nm[neg[p2]] = 1
        pos_masks.append(pm)
        neg_masks.append(nm)
    return pos_masks, neg_masks

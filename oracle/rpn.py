"""ORACLE (test infrastructure only): the RPN proper on CPU -- flatten, decode, top-k, filter, NMS,
target assignment, sampling and losses -- plus the top-level detector.

Restates reference nerf_rpn/model/rpn.py:20-27,105-130 (flatten), :240-290 (targets), :292-370
(filter), :372-456 (losses incl. the 2-D projection term :37-102), :458-536 (forward),
coder/base_bbox_coder.py:14-86, and nerf_rpn.py:129-217 (pad/stack + forward contract).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import anchors as A
from . import boxes as B
from . import coders as C
from . import geometry as G


def flatten_levels(per_level, width):
    """(N, A*width, X, Y, Z) per level -> (N, sum(X*Y*Z*A), width); rpn.py:20-27,105-130."""
    out = []
    for t in per_level:
        n, ac, x, y, z = t.shape
        out.append(t.view(n, ac // width, width, x, y, z).permute(0, 3, 4, 5, 1, 2).reshape(n, -1, width))
    return out


def _unit(v):
    n = np.linalg.norm(v)
    return v / (n if n != 0 else 1)


def view_matrices(res):
    """get_w2cs, rpn.py:37-83: four world->camera matrices looking at the grid centre."""
    ctr = np.array([res / 2] * 3)
    mats = []
    for p in np.array([[res, res, res], [res, -res, res], [-res, res, res], [-res, -res, res]]) + ctr:
        up = np.array([0, 0, 1])
        zax = _unit(p - ctr)
        xax = _unit(np.cross(up, zax))
        yax = _unit(np.cross(zax, xax))
        c2w = np.eye(4)
        c2w[:3, :3] = np.stack([xax, yax, zax], axis=1)
        c2w[:3, 3] = p
        mats.append(torch.Tensor(np.linalg.inv(c2w)))
    return mats


def projection_loss(pred_boxes, target_boxes, pos, max_dim):
    """rpn.py:421-453."""
    K = torch.tensor([[600., 0., 320.], [0., 600., 240.], [0., 0., 1.]])
    if target_boxes.shape[1] == 6:
        p = torch.cat([pred_boxes[pos, :3], pred_boxes[pos, 3:]], dim=0)
        t = torch.cat([target_boxes[pos, :3], target_boxes[pos, 3:]], dim=0)
    else:
        p = C.obb3d_extreme_points(pred_boxes[pos])
        t = C.obb3d_extreme_points(target_boxes[pos])
    p = torch.cat([p, torch.ones(p.shape[0], 1)], dim=1)
    t = torch.cat([t, torch.ones(t.shape[0], 1)], dim=1)

    def proj(M, pts):
        cam = M @ pts.t().float()
        pic = K @ cam[:3]
        return (pic[:2] / pic[2]).t()

    ps, ts = [], []
    for M in view_matrices(max_dim):
        ps.append(proj(M, p))
        ts.append(proj(M, t))
    return F.smooth_l1_loss(torch.cat(ps), torch.cat(ts), beta=1 / 9, reduction="sum") / pos.numel() / max_dim


def rotated_iou_loss(pred, target, kind):
    """RotatedIOULoss, rpn.py:133-164 (weight=None)."""
    pred, target = pred.unsqueeze(0), target.unsqueeze(0)
    if kind in ("iou", "linear_iou"):
        iou, _, _, _, u = G.iou_3d(pred, target, verbose=True)
        iou = (iou * u + 1.0) / (u + 1.0)
        loss = -torch.log(iou) if kind == "iou" else 1 - iou
    elif kind == "giou":
        loss = G.giou_3d(pred, target)[0]
    elif kind == "diou":
        loss = G.diou_3d(pred, target)[0]
    else:
        raise NotImplementedError(kind)
    return loss.sum()


class RPN:
    """RegionProposalNetwork (rpn.py:167-536) without the nn.Module trappings."""

    def __init__(self, head, pre_nms_top_n=2500, post_nms_top_n=2500, nms_thresh=0.3, fg_iou=0.35, bg_iou=0.2,
                 per_scene=256, pos_fraction=0.5, score_thresh=0.0, iou_chunk=16, rotated=False,
                 reg_loss_type="smooth_l1", sizes=A.SIZES):
        self.head, self.rotated = head, rotated
        self.pre, self.post, self.nms_thresh = pre_nms_top_n, post_nms_top_n, nms_thresh
        self.fg, self.bg, self.per_scene, self.pos_fraction = fg_iou, bg_iou, per_scene, pos_fraction
        self.score_thresh, self.iou_chunk, self.reg_loss_type = score_thresh, iou_chunk, reg_loss_type
        self.sizes = sizes
        self.box_w = 7 if rotated else 6
        self.delta_w = 8 if rotated else 6
        self.min_size = 1e-3
        self.sampler_hook = None  # tests may inject fixed (pos_idx, neg_idx)

    def decode(self, deltas, anchors):
        return C.midpoint_decode(deltas, anchors) if self.rotated else C.aabb_decode(deltas, anchors)

    def encode(self, gt, anchors):
        return C.midpoint_encode(gt, anchors) if self.rotated else C.aabb_encode(gt, anchors)

    # ---- eval ------------------------------------------------------------------------------
    def filter(self, boxes, logits, level_of, grid_sizes, per_level_counts, pad_mask):
        """rpn.py:303-370.  boxes [N,T,w], logits [N,T] -> per-scene (boxes, scores, levels)."""
        logits = logits.detach().clone()
        if pad_mask is not None:
            logits[~pad_mask] = -torch.inf
        top, off = [], 0
        for chunk in logits.split(per_level_counts, 1):
            k = min(self.pre, chunk.shape[1])
            top.append(chunk.topk(k, dim=1)[1] + off)
            off += chunk.shape[1]
        top = torch.cat(top, dim=1)
        rows = torch.arange(logits.shape[0])[:, None]
        prob = torch.sigmoid(logits[rows, top])
        lv = level_of[None, :].expand_as(logits)[rows, top]
        bx = boxes[rows, top]
        out_b, out_s, out_l = [], [], []
        for b, s, l, size in zip(bx, prob, lv, grid_sizes):
            b = B.clip_to_grid(b, size)
            idx_lvl = l.to(b.dtype)
            keep = B.big_enough(b, self.min_size)
            b, s, l, idx_lvl = b[keep], s[keep], l[keep], idx_lvl[keep]
            keep = torch.where(s >= self.score_thresh)[0]
            b, s, l, idx_lvl = b[keep], s[keep], l[keep], idx_lvl[keep]
            keep = B.nms_per_level(b, s, l, self.nms_thresh)[: self.post]
            out_b.append(b[keep]); out_s.append(s[keep]); out_l.append(idx_lvl[keep])
        return out_b, out_s, out_l

    # ---- train -----------------------------------------------------------------------------
    def assign(self, anchors, targets, pad_mask):
        """rpn.py:240-290 -> labels [A] in {1,0,-1}, matched gt [A,w]."""
        labels, matched = [], []
        for i, (a, gt) in enumerate(zip(anchors, targets)):
            if gt.numel() == 0:
                m = torch.zeros(a.shape, dtype=torch.float32)
                lab = torch.zeros((a.shape[0],), dtype=torch.float32)
            else:
                q = B.iou_matrix_chunked(C.obb3d_to_hbb(gt) if gt.shape[1] == 7 else gt, a, self.iou_chunk)
                if pad_mask is not None:
                    q[:, ~pad_mask[i]] = -1.0
                idx = B.match(q, self.fg, self.bg, True)
                m = gt[idx.clamp(min=0)]
                lab = (idx >= 0).to(torch.float32)
                lab[idx == B.BELOW] = 0.0
                lab[idx == B.BETWEEN] = -1.0
            if pad_mask is not None:
                lab[~pad_mask[i]] = -1.0
            labels.append(lab)
            matched.append(m)
        return labels, matched

    def losses(self, logits, deltas, labels, reg_targets, pred_boxes, matched, max_dim):
        """rpn.py:372-456."""
        if self.sampler_hook is not None:
            pos, neg = self.sampler_hook(labels)
        else:
            pm, nm = B.sample_pos_neg(labels, self.per_scene, self.pos_fraction)
            pos = torch.where(torch.cat(pm))[0]
            neg = torch.where(torch.cat(nm))[0]
        both = torch.cat([pos, neg])
        logits = logits.flatten()
        lab = torch.cat(labels)
        tgt = torch.cat(reg_targets)
        mt = torch.cat(matched)
        if self.reg_loss_type == "smooth_l1":
            reg = F.smooth_l1_loss(deltas[pos], tgt[pos], beta=1 / 9, reduction="sum") / both.numel()
        else:
            reg = rotated_iou_loss(pred_boxes[pos], mt[pos], self.reg_loss_type) / both.numel()
        obj = F.binary_cross_entropy_with_logits(logits[both], lab[both])
        reg2d = projection_loss(pred_boxes, mt, pos, max_dim)
        return obj, reg, reg2d, dict(pos=pos, neg=neg)

    def __call__(self, meshes, feats, ori_sizes, targets=None, training=False):
        """rpn.py:458-536."""
        logits_l, deltas_l = self.head(feats)
        grids = [tuple(f.shape[-3:]) for f in feats]
        mesh_size = tuple(meshes.shape[-3:])
        n = meshes.shape[0]
        per_level = A.all_anchors(mesh_size, grids, self.sizes)
        counts = [a.shape[0] for a in per_level]
        num_a = per_level[0].shape[0] // (grids[0][0] * grids[0][1] * grids[0][2])
        flat_logits = torch.cat(flatten_levels(logits_l, 1), dim=1)           # [N,T,1]
        flat_deltas_l = flatten_levels(deltas_l, self.delta_w)
        flat_deltas = torch.cat(flat_deltas_l, dim=1)                         # [N,T,dw]
        pad = A.padding_masks(mesh_size, grids, ori_sizes, num_a) if n > 1 else None
        anchors_cat = torch.cat(per_level)
        aux = dict(anchors=anchors_cat, logits=flat_logits.reshape(n, -1), deltas=flat_deltas)
        if not training:
            boxes = torch.stack([self.decode(flat_deltas[i].detach(), anchors_cat) for i in range(n)])
            level_of = torch.cat([torch.full((c,), i, dtype=torch.int64) for i, c in enumerate(counts)])
            b, s, l = self.filter(boxes, flat_logits.reshape(n, -1), level_of, [mesh_size] * n, counts, pad)
            aux["decoded"] = boxes
            return b, l, {}, s, aux
        anchors = [anchors_cat] * n
        obj = flat_logits.reshape(-1, 1)
        dl = flat_deltas.reshape(-1, self.delta_w)
        pred = self.decode(dl, torch.cat(anchors))
        labels, matched = self.assign(anchors, targets, pad)
        reg_t = [self.encode(m, a) for m, a in zip(matched, anchors)]
        lo, lr, l2, samp = self.losses(obj, dl, labels, reg_t, pred, matched, max(mesh_size))
        aux.update(labels=labels, matched=matched, reg_targets=reg_t, sampled=samp, pred=pred)
        return None, None, dict(loss_objectness=lo, loss_rpn_box_reg=lr, loss_rpn_box_reg_2d=l2), None, aux


class Detector:
    """NeRFRegionProposalNetwork.forward contract (nerf_rpn.py:129-217)."""

    def __init__(self, backbone, rpn):
        self.backbone, self.rpn = backbone, rpn

    def __call__(self, meshes, targets=None, training=False):
        ori = [tuple(m.shape[-3:]) for m in meshes]
        if len(meshes) > 1:
            tgt = np.max([m.shape for m in meshes], axis=0)
            meshes = [F.pad(m, (0, int(tgt[-1] - m.shape[-1]), 0, int(tgt[-2] - m.shape[-2]),
                                0, int(tgt[-3] - m.shape[-3]))) for m in meshes]
        x = torch.stack(meshes)
        feats = list(self.backbone(x))
        boxes, levels, losses, scores, aux = self.rpn(x, feats, ori, targets, training)
        aux["features"] = feats
        return [feats, boxes, levels], losses, scores, aux

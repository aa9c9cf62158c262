"""ORACLE (test infrastructure only): the FCOS variant of the hot path on CPU -- head, locations, target assignment with
centre sampling, focal / IoU / centerness losses, and the post-processor.

Restates reference nerf_rpn/model/fcos/fcos.py:17-250 (Scale, FCOSHead, FCOSModule), fcos/inference.py:48-195
(FCOSPostProcessor), fcos/utils.py:12-105 (decode_fcos_obb / encode_fcos_obb) and fcos/loss.py:77-591 (IOULoss,
RotatedIOULoss, FCOSLossComputation).  Pinned to the reference by tests/golden/make_golden.py (gen_fcos).
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import boxes as B
from . import geometry as G
from .rpn import view_matrices

INF = 100000000
SIZES_OF_INTEREST = [[-1, 16], [16, 32], [32, 64], [64, INF]]       # loss.py:272-277


class Scale(nn.Module):
    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.FloatTensor([init_value]))

    def forward(self, x):
        return x * self.scale


class FCOSHead(nn.Module):
    """fcos.py:27-130; same parameter names."""

    def __init__(self, in_channels=256, num_convs=4, fpn_strides=(4, 8, 16, 32), norm_reg_targets=True, centerness_on_reg=True,
                 use_obb=False):
        super().__init__()
        self.fpn_strides, self.norm_reg_targets, self.centerness_on_reg, self.use_obb = fpn_strides, norm_reg_targets, centerness_on_reg, use_obb

        def tower():
            mods = []
            for _ in range(num_convs):
                mods += [nn.Conv3d(in_channels, in_channels, 3, padding=1), nn.GroupNorm(32, in_channels), nn.ReLU()]
            return nn.Sequential(*mods)
        self.cls_tower, self.bbox_tower = tower(), tower()
        self.cls_logits = nn.Conv3d(in_channels, 1, 3, padding=1)
        self.bbox_pred = nn.Conv3d(in_channels, 8 if use_obb else 6, 3, padding=1)
        self.centerness = nn.Conv3d(in_channels, 1, 3, padding=1)
        self.scales = nn.ModuleList(Scale(1.0) for _ in range(5))

    def forward(self, feats):
        logits, regs, ctrs = [], [], []
        for l, f in enumerate(feats):
            ct, bt = self.cls_tower(f), self.bbox_tower(f)
            logits.append(self.cls_logits(ct))
            ctrs.append(self.centerness(bt if self.centerness_on_reg else ct))
            r = self.scales[l](self.bbox_pred(bt))
            if self.norm_reg_targets:
                r = torch.cat([F.relu(r[:, :6]), r[:, 6:]], dim=1)
                if not self.training:
                    r = torch.cat([r[:, :6] * self.fpn_strides[l], r[:, 6:]], dim=1)
            else:
                r = torch.exp(r)
            regs.append(r)
        return logits, regs, ctrs


def locations_of(feats, strides):
    """compute_locations, fcos.py:221-250: arange(0, n*s, s) + s // 2, (x, y, z) ij-meshgrid order."""
    out = []
    for f, s in zip(feats, strides):
        w, l, h = f.shape[-3:]
        g = torch.meshgrid(*[torch.arange(0, n * s, step=s, dtype=torch.float32) for n in (w, l, h)], indexing="ij")
        out.append(torch.stack([t.reshape(-1) for t in g], dim=1) + s // 2)
    return out


def padding_masks(locations, sizes):
    """fcos.py:252-266."""
    return [torch.stack([(loc[:, 0] < w) & (loc[:, 1] < l) & (loc[:, 2] < h) for (w, l, h) in sizes]) for loc in locations]


def decode_obb(loc, reg):
    """decode_fcos_obb, fcos/utils.py:12-62."""
    x0, y0, z0 = loc[:, 0] - reg[:, 0], loc[:, 1] - reg[:, 1], loc[:, 2] - reg[:, 2]
    x1, y1, z1 = loc[:, 0] + reg[:, 3], loc[:, 1] + reg[:, 4], loc[:, 2] + reg[:, 5]
    vx = (x1 + x0) / 2 + reg[:, 6] * (x1 - x0)
    vy = (y1 + y0) / 2 + reg[:, 7] * (y1 - y0)
    vx = torch.clamp(vx, min=x0, max=x1)
    vy = torch.clamp(vy, min=y0, max=y1)
    ctr = torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (z0 + z1) / 2], dim=1)
    v0 = torch.stack([vx, y1], dim=1) - ctr[:, :2]
    v1 = torch.stack([x1, vy], dim=1) - ctr[:, :2]
    d0, d1 = torch.norm(v0, dim=1), torch.norm(v1, dim=1)
    dmax = torch.max(d0, d1)
    v0 = v0 / (d0[:, None] + 1e-7) * dmax[:, None] + ctr[:, :2]
    v1 = v1 / (d1[:, None] + 1e-7) * dmax[:, None] + ctr[:, :2]
    length = torch.norm(v0 - v1, dim=1)
    width = torch.norm((v0 + v1) / 2 - ctr[:, :2], dim=1) * 2
    mid = (v0 + v1) / 2 - ctr[:, :2]
    mid = torch.where(((mid[:, 0] == 0) & (mid[:, 1] == 0))[:, None], torch.tensor([1e-7, 0.0]).expand_as(mid), mid)
    return torch.stack([ctr[:, 0], ctr[:, 1], ctr[:, 2], width, length, z1 - z0, torch.atan2(mid[:, 1], mid[:, 0])], dim=1)


def obb_summary(boxes):
    """Per-GT quantities of encode_fcos_obb (fcos/utils.py:65-105) that do not depend on the location:
    AABB of the footprint + z range, and the midpoint offsets alpha, beta."""
    c = G.corners_2d(boxes[:, [0, 1, 3, 4, 6]].unsqueeze(0)).squeeze(0)      # [G,4,2]
    xs, ys = c[:, :, 0], c[:, :, 1]
    xmax, ymax, xmin, ymin = xs.max(1)[0], ys.max(1)[0], xs.min(1)[0], ys.min(1)[0]
    xt, yt = xs.clone(), ys.clone()
    xt[ymax.unsqueeze(1) - ys > 0.1] = -1e6
    yt[xmax.unsqueeze(1) - xs > 0.1] = 1e6
    vx, vy = xt.max(1)[0], yt.min(1)[0]
    ids = torch.isclose(vx, xmax) & torch.isclose(vy, ymin)
    vx = torch.where(ids, xmax, vx)
    vy = torch.where(ids, ymin, vy)
    alpha = (vx - boxes[:, 0]) / (xmax - xmin)
    beta = (vy - boxes[:, 1]) / (ymax - ymin)
    aabb = torch.stack([xmin, ymin, boxes[:, 2] - boxes[:, 5] / 2, xmax, ymax, boxes[:, 2] + boxes[:, 5] / 2], dim=1)
    return aabb, alpha, beta


def sample_region(gt, strides, counts, loc, radius):
    """get_sample_region, loss.py:209-268: location strictly inside the GT box shrunk to centre +- stride*radius."""
    K, n = loc.shape[0], gt.shape[0]
    g = gt[None].expand(K, n, 6)
    ctr = (g[..., :3] + g[..., 3:]) / 2
    r = torch.cat([torch.full((c,), s * radius) for c, s in zip(counts, strides)])[:, None, None]
    lo = torch.where(ctr - r > g[..., :3], ctr - r, g[..., :3])
    hi = torch.where(ctr + r > g[..., 3:], g[..., 3:], ctr + r)
    d = torch.cat([loc[:, None, :] - lo, hi - loc[:, None, :]], dim=-1)
    return d.min(-1)[0] > 0


def targets_for_scene(loc, counts, strides, gt, radius, use_obb):
    """compute_targets_for_locations[_obb], loss.py:318-437 -> labels [K], reg_targets [K, 6|8] (not yet stride-normalised)."""
    D = 8 if use_obb else 6
    if gt.shape[0] == 0:
        return torch.zeros(loc.shape[0]), torch.zeros(loc.shape[0], D)
    if use_obb:
        aabb, alpha, beta = obb_summary(gt)
    else:
        aabb = gt
    reg = torch.cat([loc[:, None, :] - aabb[None, :, :3], aabb[None, :, 3:] - loc[:, None, :]], dim=2)      # [K,G,6]
    if use_obb:
        reg = torch.cat([reg, alpha[None, :, None].expand(loc.shape[0], -1, 1), beta[None, :, None].expand(loc.shape[0], -1, 1)], dim=2)
    inside = sample_region(aabb, strides, counts, loc, radius) if radius > 0 else reg[..., :6].min(2)[0] > 0
    soi = torch.cat([torch.tensor(SIZES_OF_INTEREST[l], dtype=torch.float32)[None].expand(c, -1) for l, c in enumerate(counts)])
    mx = reg[..., :6].max(2)[0]
    cared = (mx >= soi[:, [0]]) & (mx <= soi[:, [1]])
    vol = ((aabb[:, 3] - aabb[:, 0]) * (aabb[:, 4] - aabb[:, 1]) * (aabb[:, 5] - aabb[:, 2]))[None].repeat(loc.shape[0], 1)
    vol[inside == 0] = INF
    vol[cared == 0] = INF
    best, which = vol.min(dim=1)
    labels = torch.ones(loc.shape[0])
    labels[best == INF] = 0
    return labels, reg[torch.arange(loc.shape[0]), which]


def centerness_targets(t):
    """loss.py:439-446."""
    lr, tb, fb = t[:, [0, 3]], t[:, [1, 4]], t[:, [2, 5]]
    return torch.sqrt((lr.min(-1)[0] / lr.max(-1)[0]) * (tb.min(-1)[0] / tb.max(-1)[0]) * (fb.min(-1)[0] / fb.max(-1)[0]))


def focal_loss_sum(logits, targets, alpha=0.25, gamma=2.0):
    """torchvision.ops.sigmoid_focal_loss(reduction='sum')."""
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    pt = p * targets + (1 - p) * (1 - targets)
    return ((alpha * targets + (1 - alpha) * (1 - targets)) * ce * (1 - pt) ** gamma).sum()


def aabb_iou_loss(pred, target, weight, kind):
    """IOULoss, loss.py:77-134 (argument order l,t,f,r,b,ba)."""
    tv = (target[:, 0] + target[:, 3]) * (target[:, 1] + target[:, 4]) * (target[:, 2] + target[:, 5])
    pv = (pred[:, 0] + pred[:, 3]) * (pred[:, 1] + pred[:, 4]) * (pred[:, 2] + pred[:, 5])
    wi = torch.min(pred[:, 0], target[:, 0]) + torch.min(pred[:, 3], target[:, 3])
    gw = torch.max(pred[:, 0], target[:, 0]) + torch.max(pred[:, 3], target[:, 3])
    hi = torch.min(pred[:, 4], target[:, 4]) + torch.min(pred[:, 1], target[:, 1])
    gh = torch.max(pred[:, 4], target[:, 4]) + torch.max(pred[:, 1], target[:, 1])
    di = torch.min(pred[:, 2], target[:, 2]) + torch.min(pred[:, 5], target[:, 5])
    gd = torch.max(pred[:, 2], target[:, 2]) + torch.max(pred[:, 5], target[:, 5])
    ac = gw * gh * gd + 1e-7
    inter = wi * hi * di
    union = tv + pv - inter
    iou = (inter + 1.0) / (union + 1.0)
    giou = iou - (ac - union) / ac
    loss = {"iou": -torch.log(iou), "linear_iou": 1 - iou, "giou": 1 - giou}[kind]
    return (loss * weight).sum() if weight.sum() > 0 else loss.sum()


def obb_iou_loss(pred, target, weight, kind):
    """RotatedIOULoss, loss.py:137-173."""
    zero = torch.zeros(pred.shape[0], 3)
    pb, tb = decode_obb(zero, pred).unsqueeze(0), decode_obb(zero, target).unsqueeze(0)
    if kind in ("iou", "linear_iou"):
        iou, _, _, _, u = G.iou_3d(pb, tb, verbose=True)
        iou = (iou * u + 1.0) / (u + 1.0)
        loss = -torch.log(iou) if kind == "iou" else 1 - iou
    elif kind == "giou":
        loss = G.giou_3d(pb, tb)[0]
    elif kind == "diou":
        loss = G.diou_3d(pb, tb)[0]
    else:
        raise NotImplementedError(kind)
    return (loss * weight).sum() if weight.sum() > 0 else loss.sum()


def projection_loss(box_reg, reg_targets, weights):
    """compute_2d_projection_loss, loss.py:448-481."""
    K = torch.tensor([[600., 0., 320.], [0., 600., 240.], [0., 0., 1.]])
    zero = torch.zeros(box_reg.shape[0], 3)

    def pts(b):      # obb2points_3d, fcos/utils.py:367-373
        ctr, w, l, h, th = torch.split(b, [3, 1, 1, 1, 1], dim=-1)
        v = torch.cat([w / 2 * torch.cos(th) - l / 2 * torch.sin(th), w / 2 * torch.sin(th) + l / 2 * torch.cos(th), h / 2], dim=-1)
        p = torch.cat([ctr - v, ctr + v], dim=0)
        return torch.cat([p, torch.ones(p.shape[0], 1)], dim=1)
    p, t = pts(decode_obb(zero, box_reg)), pts(decode_obb(zero, reg_targets))

    def proj(M, q):
        pic = K @ (M @ q.t().float())[:3]
        return (pic[:2] / pic[2]).t()
    mats = view_matrices(160)
    p2, t2 = torch.cat([proj(M, p) for M in mats]), torch.cat([proj(M, t) for M in mats])
    loss = F.smooth_l1_loss(p2, t2, beta=1 / 9, reduction="none") / 160
    factor = loss.shape[0] // weights.shape[0]
    return (loss * weights[:, None].repeat(factor, 1)).sum() / (factor * loss.shape[1])


class FCOS:
    """FCOSModule + FCOSOverNeRF (fcos.py:133-386) without the nn.Module trappings; single process (world_size 1)."""

    def __init__(self, backbone, head, strides=(4, 8, 16, 32), use_obb=False, center_sampling_radius=1.5, iou_loss_type="iou",
                 norm_reg_targets=True, use_additional_l1_loss=False, proj2d_loss_weight=0.0, pre_nms_thresh=0.0, pre_nms_top_n=2500,
                 nms_thresh=0.3, fpn_post_nms_top_n=2500, min_size=0.0):
        self.backbone, self.head, self.strides, self.use_obb = backbone, head, list(strides), use_obb
        self.radius, self.iou_loss_type, self.norm_reg_targets = center_sampling_radius, iou_loss_type, norm_reg_targets
        self.use_additional_l1_loss, self.proj2d_loss_weight = use_additional_l1_loss, proj2d_loss_weight
        self.pre_nms_thresh, self.pre_nms_top_n, self.nms_thresh = pre_nms_thresh, pre_nms_top_n, nms_thresh
        self.fpn_post_nms_top_n, self.min_size = fpn_post_nms_top_n, min_size

    # ---- training (loss.py:483-591)
    def losses(self, locations, box_cls, box_reg, ctr, targets, masks):
        D = 8 if self.use_obb else 6
        counts = [len(p) for p in locations]
        allp = torch.cat(locations)
        per_scene = [targets_for_scene(allp, counts, self.strides, gt, self.radius, self.use_obb) for gt in targets]
        labels, regt = [], []
        for l in range(len(locations)):         # level-first, scenes inside (loss.py:298-314)
            lab = torch.cat([torch.split(ls, counts)[l] for ls, _ in per_scene])
            rt = torch.cat([torch.split(rs, counts)[l] for _, rs in per_scene]).clone()
            if self.norm_reg_targets:
                rt[:, :6] = rt[:, :6] / self.strides[l]
            labels.append(lab)
            regt.append(rt)
        cls_f = torch.cat([c.permute(0, 2, 3, 4, 1).reshape(-1) for c in box_cls])
        reg_f = torch.cat([r.permute(0, 2, 3, 4, 1).reshape(-1, D) for r in box_reg])
        ctr_f = torch.cat([c.reshape(-1) for c in ctr])
        lab_f, rt_f = torch.cat(labels), torch.cat(regt)
        if masks is not None:
            m = torch.cat([mk.reshape(-1) for mk in masks])
            cls_f, reg_f, ctr_f, lab_f, rt_f = cls_f[m], reg_f[m], ctr_f[m], lab_f[m], rt_f[m]
        pos = torch.nonzero(lab_f > 0).squeeze(1)
        reg_p, rt_p, ctr_p = reg_f[pos], rt_f[pos], ctr_f[pos]
        npos = max(float(pos.numel()), 1.0)
        cls_loss = focal_loss_sum(cls_f, lab_f) / npos
        aux = {"labels": lab_f, "reg_targets": rt_f, "pos": pos}
        if pos.numel() == 0:
            return cls_loss, reg_p.sum(), ctr_p.sum(), aux
        ct = centerness_targets(rt_p)
        norm = ct.sum().item()
        if self.iou_loss_type != "smooth_l1":
            fn = obb_iou_loss if self.use_obb else aabb_iou_loss
            reg_loss = fn(reg_p, rt_p, ct, self.iou_loss_type) / norm
        else:
            reg_loss = (F.smooth_l1_loss(reg_p, rt_p, reduction="none") * ct.unsqueeze(1)).sum() / norm
        ctr_loss = F.binary_cross_entropy_with_logits(ctr_p, ct, reduction="sum") / npos
        if self.use_obb and self.use_additional_l1_loss and self.iou_loss_type != "smooth_l1":
            reg_loss = reg_loss + (F.smooth_l1_loss(reg_p[:, 6:], rt_p[:, 6:], reduction="none") * ct.unsqueeze(-1)).sum() / norm
        if self.use_obb and self.proj2d_loss_weight > 0:
            reg_loss = reg_loss + projection_loss(reg_p, rt_p, ct) / norm * self.proj2d_loss_weight
        aux["centerness_targets"] = ct
        return cls_loss, reg_loss, ctr_loss, aux

    # ---- inference (inference.py:48-195)
    def select(self, locations, box_cls, box_reg, ctr, sizes, masks):
        N = box_cls[0].shape[0]
        D = 8 if self.use_obb else 6
        boxes_all, scores_all, levels_all = [[] for _ in range(N)], [[] for _ in range(N)], [[] for _ in range(N)]
        for lvl, (loc, c, r, t) in enumerate(zip(locations, box_cls, box_reg, ctr)):
            c = c.permute(0, 2, 3, 4, 1).reshape(N, -1, 1).sigmoid()
            r = r.permute(0, 2, 3, 4, 1).reshape(N, -1, D)
            t = t.permute(0, 2, 3, 4, 1).reshape(N, -1).sigmoid()
            if masks is not None:
                c[~masks[lvl]] = -1e5
            cand = c > self.pre_nms_thresh
            top_n = cand.view(N, -1).sum(1).clamp(max=self.pre_nms_top_n)
            c = c * t[:, :, None]
            for i in range(N):
                s = c[i][cand[i]]
                where = cand[i].nonzero()[:, 0]
                rr, ll = r[i][where], loc[where]
                if cand[i].sum().item() > top_n[i].item():
                    s, k = s.topk(int(top_n[i]), sorted=False)
                    rr, ll = rr[k], ll[k]
                if not self.use_obb:
                    det = torch.cat([ll - rr[:, :3], ll + rr[:, 3:6]], dim=1)
                    det = B.clip_to_grid(det, sizes[i])
                else:
                    det = decode_obb(ll, rr)
                keep = B.big_enough(det, self.min_size)
                boxes_all[i].append(det[keep])
                scores_all[i].append(torch.sqrt(s[keep]))
                levels_all[i].append(torch.full((keep.numel(),), float(lvl)))
        out_b, out_s = [], []
        for i in range(N):
            b, s, lv = torch.cat(boxes_all[i]), torch.cat(scores_all[i]), torch.cat(levels_all[i])
            keep = B.greedy_nms(b, s, self.nms_thresh)
            b, s, lv = b[keep], s[keep], lv[keep]
            if keep.numel() > self.fpn_post_nms_top_n > 0:
                thr, _ = torch.kthvalue(s, keep.numel() - self.fpn_post_nms_top_n + 1)
                k2 = torch.nonzero(s >= thr.item()).squeeze(1)
                b, s, lv = b[k2], s[k2], lv[k2]
            out_b.append(torch.cat([lv[:, None], b], dim=1))
            out_s.append(s)
        return out_b, out_s

    def __call__(self, meshes, targets=None, training=False):
        sizes = [tuple(m.shape[-3:]) for m in meshes]
        tgt = [max(s[d] for s in sizes) for d in range(3)]
        x = torch.stack([F.pad(m, (0, tgt[2] - m.shape[-1], 0, tgt[1] - m.shape[-2], 0, tgt[0] - m.shape[-3])) for m in meshes])
        feats = list(self.backbone(x))
        self.head.train(training)
        box_cls, box_reg, ctr = self.head(feats)
        locations = locations_of(feats, self.strides)
        masks = padding_masks(locations, sizes) if len(meshes) > 1 else None
        if training:
            cl, rl, tl, aux = self.losses(locations, box_cls, box_reg, ctr, targets, masks)
            return None, {"loss_cls": cl, "loss_reg": rl, "loss_centerness": tl}, None, aux
        boxes, scores = self.select(locations, box_cls, box_reg, ctr, sizes, masks)
        return boxes, {}, scores, {"feats": feats, "box_cls": box_cls, "box_reg": box_reg, "centerness": ctr}

"""ORACLE (test infrastructure only): recall / AP of box proposals on CPU.
Restates reference nerf_rpn/eval.py:14-81 (recall, greedy GT<->proposal matching) and :319-395 (VOC AP)."""
import torch

from . import boxes as B


def recall(props_list, scores_list, gt_list, thresholds=None, limit=None):
    cov, npos = [], 0
    for p, s, g in zip(props_list, scores_list, gt_list):
        p = p[torch.argsort(s, descending=True)]
        if p.shape[0] == 0 or g.shape[0] == 0:
            continue
        npos += g.shape[0]
        if limit is not None and len(p) > limit:
            p = p[:limit]
        ov = B.iou_matrix(p, g)
        c = torch.zeros(g.shape[0])
        for j in range(min(p.shape[0], g.shape[0])):
            mo, am = ov.max(dim=0)
            go, gi = mo.max(dim=0)
            bi = am[gi]
            c[j] = ov[bi, gi]
            ov[bi, :] = -1
            ov[:, gi] = -1
        cov.append(c)
    cov = torch.sort(torch.cat(cov) if cov else torch.zeros(0))[0]
    if thresholds is None:
        thresholds = torch.arange(0.5, 0.95 + 1e-5, 0.05, dtype=torch.float32)
    rec = torch.stack([(cov >= t).float().sum() / float(npos) for t in thresholds])
    return {"ar": rec.mean(), "recalls": rec, "num_pos": npos, "gt_overlaps": cov}


def average_precision(props_list, scores_list, gt_list, iou_thresh=0.25, top_k=None):
    ngt, sid, dets, scs = 0, [], [], []
    for i, (p, s, g) in enumerate(zip(props_list, scores_list, gt_list)):
        if top_k is not None and len(p) > top_k:
            ids = torch.argsort(s, descending=True)[:top_k]
            p, s = p[ids], s[ids]
        sid += [i] * len(p)
        dets.append(p)
        scs.append(s)
        ngt += g.shape[0]
    sid, dets, scs = torch.tensor(sid, dtype=torch.int64), torch.cat(dets), torch.cat(scs)
    order = torch.argsort(scs, descending=True)
    dets, sid = dets[order], sid[order]
    used = [torch.zeros(len(g), dtype=torch.bool) for g in gt_list]
    tp = torch.zeros(len(dets), dtype=torch.bool)
    fp = torch.zeros(len(dets), dtype=torch.bool)
    for i in range(len(dets)):
        ov = B.iou_matrix(dets[i].unsqueeze(0), gt_list[sid[i]])
        m, a = ov.max(dim=1)
        if m > iou_thresh and not used[sid[i]][a]:
            tp[i] = True
            used[sid[i]][a] = True
        else:
            fp[i] = True
    tp, fp = torch.cumsum(tp, 0), torch.cumsum(fp, 0)
    rec, pre = tp / ngt, tp / (tp + fp)
    mrec = torch.cat((torch.tensor([0.0]), rec, torch.tensor([1.0])))
    mpre = torch.cat((torch.tensor([0.0]), pre, torch.tensor([0.0])))
    for i in range(mpre.size(0) - 1, 0, -1):
        mpre[i - 1] = torch.max(mpre[i - 1], mpre[i])
    idx = torch.where(mrec[1:] != mrec[:-1])[0]
    return {"ap": torch.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1])}

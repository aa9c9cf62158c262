"""Rotated 3D IoU / GIoU / DIoU with the reference's function names (rotated_iou/oriented_iou_loss.py:82-148).

Two paths:
  * no gradient needed (NMS, metrics, matching): one fused HIP kernel per pair (``ops.iou3d_pair``);
  * gradient needed (IoU-type regression losses on the <= 128 sampled positives per scene): the polygon-clipping
    formulation in torch ops with the vertex sort done by the HIP drop-in of the reference's ``sort_vertices`` op --
    the same split the reference uses (everything differentiable except the sort)."""
import torch

from ... import ops

_SX = (0.5, -0.5, -0.5, 0.5)
_SY = (0.5, 0.5, -0.5, -0.5)


def box2corners_th(box):
    x, y, w, h, a = [box[..., i:i + 1] for i in range(5)]
    sx = torch.tensor(_SX, device=box.device)
    sy = torch.tensor(_SY, device=box.device)
    lx, ly = sx * w, sy * h
    s, c = torch.sin(a), torch.cos(a)
    return torch.stack([lx * c - ly * s + x, lx * s + ly * c + y], dim=-1)


def _intersection_area(c1, c2):
    B, N = c1.shape[:2]
    n1, n2 = c1[:, :, [1, 2, 3, 0], :], c2[:, :, [1, 2, 3, 0], :]
    x1, y1 = c1[..., 0][..., :, None], c1[..., 1][..., :, None]
    x2, y2 = n1[..., 0][..., :, None], n1[..., 1][..., :, None]
    x3, y3 = c2[..., 0][..., None, :], c2[..., 1][..., None, :]
    x4, y4 = n2[..., 0][..., None, :], n2[..., 1][..., None, :]
    num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)
    den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)
    with torch.no_grad():
        par = num == 0.0
        t = torch.where(par, torch.full_like(num, -1.0), den_t / num)
        u = torch.where(par, torch.full_like(num, -1.0), -den_u / num)
        ok = (t > 0) & (t < 1) & (u > 0) & (u < 1)
    ts = den_t / (num + 1e-8)
    pts = torch.stack([x1 + ts * (x2 - x1), y1 + ts * (y2 - y1)], dim=-1) * ok.float().unsqueeze(-1)

    def inside(p, q):
        a, b, d = q[:, :, 0:1, :], q[:, :, 1:2, :], q[:, :, 3:4, :]
        ab, ad, am = b - a, d - a, p - a
        r1 = (ab * am).sum(-1) / (ab * ab).sum(-1)
        r2 = (ad * am).sum(-1) / (ad * ad).sum(-1)
        return (r1 > -1e-6) & (r1 < 1 + 1e-6) & (r2 > -1e-6) & (r2 < 1 + 1e-6)

    verts = torch.cat([c1, c2, pts.reshape(B, N, 16, 2)], dim=2)
    mask = torch.cat([inside(c1, c2), inside(c2, c1), ok.reshape(B, N, 16)], dim=2)
    nv = mask.int().sum(dim=2).int()
    mean = (verts * mask.float().unsqueeze(-1)).sum(dim=2, keepdim=True) / nv[..., None, None]
    order = ops.sort_vertices((verts - mean).detach(), mask, nv).long()
    sel = torch.gather(verts, 2, order.unsqueeze(-1).expand(-1, -1, -1, 2))
    cross = sel[:, :, :-1, 0] * sel[:, :, 1:, 1] - sel[:, :, :-1, 1] * sel[:, :, 1:, 0]
    return cross.sum(dim=2).abs() / 2


def cal_iou(box1, box2):
    c1, c2 = box2corners_th(box1), box2corners_th(box2)
    inter = _intersection_area(c1, c2)
    u = box1[..., 2] * box1[..., 3] + box2[..., 2] * box2[..., 3] - inter
    return inter / u, c1, c2, u


def cal_iou_3d(box3d1, box3d2, verbose=False):
    if not verbose and not (box3d1.requires_grad or box3d2.requires_grad):
        return ops.iou3d_pair(box3d1, box3d2)
    sel = [0, 1, 3, 4, 6]
    zt1, zb1 = box3d1[..., 2] + box3d1[..., 5] * 0.5, box3d1[..., 2] - box3d1[..., 5] * 0.5
    zt2, zb2 = box3d2[..., 2] + box3d2[..., 5] * 0.5, box3d2[..., 2] - box3d2[..., 5] * 0.5
    zov = (torch.min(zt1, zt2) - torch.max(zb1, zb2)).clamp_min(0.)
    iou2, c1, c2, u = cal_iou(box3d1[..., sel], box3d2[..., sel])
    inter = iou2 * u * zov
    u3 = box3d1[..., 3] * box3d1[..., 4] * box3d1[..., 5] + box3d2[..., 3] * box3d2[..., 4] * box3d2[..., 5] - inter
    if verbose:
        zr = (torch.max(zt1, zt2) - torch.min(zb1, zb2)).clamp_min(0.)
        return inter / u3, c1, c2, zr, u3
    return inter / u3


def _hull_tables(device):
    skip = {(0, 2), (1, 3), (5, 7), (4, 6)}
    lines, rest = [], []
    for i in range(8):
        for j in range(i + 1, 8):
            if (i, j) not in skip:
                lines.append([i, j])
                rest.append([k for k in range(8) if k not in (i, j)])
    return torch.tensor(lines, device=device), torch.tensor(rest, device=device)


def smallest_bounding_box(c8):
    lines, rest = _hull_tables(c8.device)
    ln, pt = c8[..., lines, :], c8[..., rest, :]
    x1, y1, x2, y2 = ln[..., 0:1, 0], ln[..., 0:1, 1], ln[..., 1:2, 0], ln[..., 1:2, 1]
    k = (y2 - y1) / (x2 - x1 + 1e-8)
    vec = torch.cat([torch.ones_like(k), k], dim=-1).unsqueeze(-2)
    proj = (torch.cat([ln, pt], dim=-2) * vec).sum(-1) / torch.norm(vec, dim=-1)
    prange = proj.max(-1)[0] - proj.min(-1)[0]
    x, y = pt[..., 0], pt[..., 1]
    d = ((y2 - y1) * x - (x2 - x1) * y + x2 * y1 - y2 * x1) / torch.sqrt((y2 - y1).square() + (x2 - x1).square() + 1e-14)
    drange = torch.max(d.max(-1)[0] - d.min(-1)[0], d.abs().max(-1)[0])
    area = prange * drange
    area = area + (area == 0).to(c8.dtype) * 1e8
    idx = area.min(dim=-1, keepdim=True)[1]
    return prange.gather(-1, idx).squeeze(-1).float(), drange.gather(-1, idx).squeeze(-1).float()


def cal_giou_3d(box3d1, box3d2, enclosing_type="smallest"):
    iou, c1, c2, zr, u3 = cal_iou_3d(box3d1, box3d2, verbose=True)
    w, h = smallest_bounding_box(torch.cat([c1, c2], dim=-2))
    vc = zr * w * h
    loss = 1. - iou + (vc - u3) / vc
    return loss, 1 - loss, iou


def cal_diou_3d(box3d1, box3d2, enclosing_type="smallest"):
    iou, c1, c2, zr, u3 = cal_iou_3d(box3d1, box3d2, verbose=True)
    w, h = smallest_bounding_box(torch.cat([c1, c2], dim=-2))
    dx, dy, dz = box3d1[..., 0] - box3d2[..., 0], box3d1[..., 1] - box3d2[..., 1], box3d1[..., 2] - box3d2[..., 2]
    return 1. - iou + (dx * dx + dy * dy + dz * dz) / (w * w + h * h + zr * zr), iou

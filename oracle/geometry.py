"""ORACLE (test infrastructure only): rotated 2D/3D box geometry on CPU.

Restates reference nerf_rpn/model/rotated_iou/oriented_iou_loss.py:6-148,
box_intersection_2d.py:11-176, min_enclosing_box.py:26-166, and the vertex sort through
``sortv.c`` (cuda_op/sort_vert_kernel.cu).  Shapes follow the reference: boxes are [B, N, 7]
(x, y, z, w, h, d, theta) *paired* element-wise, not all-pairs.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_sortv.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(so)
        _LIB.oracle_sort_vertices.argtypes = [ctypes.c_int64, ctypes.c_int] + [ctypes.c_void_p] * 4
        _LIB.oracle_sort_vertices.restype = None
    return _LIB


def sort_vertices(vertices, mask, num_valid):
    """f32[B,N,M,2], bool[B,N,M], i32[B,N] -> i32[B,N,9]  (sort_vert.cpp:6-33)."""
    v = np.ascontiguousarray(vertices.detach().cpu().numpy(), dtype=np.float32)
    m = np.ascontiguousarray(mask.detach().cpu().numpy().astype(np.uint8))
    nv = np.ascontiguousarray(num_valid.detach().cpu().numpy(), dtype=np.int32)
    B, N, M = m.shape
    out = np.zeros((B, N, 9), dtype=np.int32)
    _lib().oracle_sort_vertices(B * N, M, v.ctypes.data, m.ctypes.data, nv.ctypes.data, out.ctypes.data)
    return torch.from_numpy(out)


_SX = torch.tensor([0.5, -0.5, -0.5, 0.5])
_SY = torch.tensor([0.5, 0.5, -0.5, -0.5])


def corners_2d(box):
    """[B,N,5] (x,y,w,h,a) -> [B,N,4,2]; oriented_iou_loss.py:6-35."""
    x, y, w, h, a = [box[..., i:i + 1] for i in range(5)]
    lx = _SX * w
    ly = _SY * h
    s, c = torch.sin(a), torch.cos(a)
    # [lx, ly] @ [[c, s], [-s, c]]
    rx = lx * c + ly * (-s)
    ry = lx * s + ly * c
    return torch.stack([rx + x, ry + y], dim=-1)


def _edge_intersections(c1, c2):
    """box_intersection_2d.py:11-52 -> points [B,N,4,4,2], mask [B,N,4,4]."""
    n1 = c1[:, :, [1, 2, 3, 0], :]
    n2 = c2[:, :, [1, 2, 3, 0], :]
    x1, y1 = c1[..., 0][..., :, None], c1[..., 1][..., :, None]
    x2, y2 = n1[..., 0][..., :, None], n1[..., 1][..., :, None]
    x3, y3 = c2[..., 0][..., None, :], c2[..., 1][..., None, :]
    x4, y4 = n2[..., 0][..., None, :], n2[..., 1][..., None, :]
    num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)
    den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)
    par = num == 0.0
    t = torch.where(par, torch.full_like(num, -1.0), den_t / num)
    u = torch.where(par, torch.full_like(num, -1.0), -den_u / num)
    ok = (t > 0) & (t < 1) & (u > 0) & (u < 1)
    ts = den_t / (num + 1e-8)
    pts = torch.stack([x1 + ts * (x2 - x1), y1 + ts * (y2 - y1)], dim=-1)
    return pts * ok.float().unsqueeze(-1), ok


def _inside(c1, c2):
    """corners of c1 inside c2; box_intersection_2d.py:54-79."""
    a, b, d = c2[:, :, 0:1, :], c2[:, :, 1:2, :], c2[:, :, 3:4, :]
    ab, ad, am = b - a, d - a, c1 - a
    pab = (ab * am).sum(-1) / (ab * ab).sum(-1)
    pad = (ad * am).sum(-1) / (ad * ad).sum(-1)
    return (pab > -1e-6) & (pab < 1 + 1e-6) & (pad > -1e-6) & (pad < 1 + 1e-6)


def intersection_area_2d(c1, c2, return_debug=False):
    """oriented_box_intersection_2d, box_intersection_2d.py:96-176."""
    B, N = c1.shape[:2]
    pts, ok = _edge_intersections(c1, c2)
    in12, in21 = _inside(c1, c2), _inside(c2, c1)
    verts = torch.cat([c1, c2, pts.reshape(B, N, 16, 2)], dim=2)
    mask = torch.cat([in12, in21, ok.reshape(B, N, 16)], dim=2)
    nv = mask.int().sum(dim=2).int()
    mean = (verts * mask.float().unsqueeze(-1)).sum(dim=2, keepdim=True) / nv[..., None, None]
    order = sort_vertices((verts - mean).float(), mask, nv).long()
    sel = torch.gather(verts, 2, order.unsqueeze(-1).expand(-1, -1, -1, 2))
    cross = sel[:, :, :-1, 0] * sel[:, :, 1:, 1] - sel[:, :, :-1, 1] * sel[:, :, 1:, 0]
    area = cross.sum(dim=2).abs() / 2
    if return_debug:
        return area, dict(vertices=verts, mask=mask, num_valid=nv, order=order)
    return area


def iou_2d(b1, b2):
    """cal_iou, oriented_iou_loss.py:37-57 -> iou, corners1, corners2, union."""
    c1, c2 = corners_2d(b1), corners_2d(b2)
    inter = intersection_area_2d(c1, c2)
    union = b1[..., 2] * b1[..., 3] + b2[..., 2] * b2[..., 3] - inter
    return inter / union, c1, c2, union


def iou_3d(b1, b2, verbose=False):
    """cal_iou_3d, oriented_iou_loss.py:82-107."""
    sel = [0, 1, 3, 4, 6]
    zt1, zb1 = b1[..., 2] + b1[..., 5] * 0.5, b1[..., 2] - b1[..., 5] * 0.5
    zt2, zb2 = b2[..., 2] + b2[..., 5] * 0.5, b2[..., 2] - b2[..., 5] * 0.5
    zov = (torch.min(zt1, zt2) - torch.max(zb1, zb2)).clamp_min(0.0)
    i2, c1, c2, u2 = iou_2d(b1[..., sel], b2[..., sel])
    inter = i2 * u2 * zov
    u3 = b1[..., 3] * b1[..., 4] * b1[..., 5] + b2[..., 3] * b2[..., 4] * b2[..., 5] - inter
    if verbose:
        zr = (torch.max(zt1, zt2) - torch.min(zb1, zb2)).clamp_min(0.0)
        return inter / u3, c1, c2, zr, u3
    return inter / u3


def _hull_tables():
    """min_enclosing_box.py:26-52: 24 candidate hull edges and the other 6 points."""
    skip = {(0, 2), (1, 3), (5, 7), (4, 6)}
    lines, rest = [], []
    for i in range(8):
        for j in range(i + 1, 8):
            if (i, j) in skip:
                continue
            lines.append([i, j])
            rest.append([k for k in range(8) if k not in (i, j)])
    return torch.tensor(lines), torch.tensor(rest)


_LINES, _REST = _hull_tables()


def enclosing_box_wh(c8):
    """smallest_bounding_box, min_enclosing_box.py:54-166.  c8: [...,8,2] -> w, h [...]."""
    ln = c8[..., _LINES, :]          # [...,24,2,2]
    pt = c8[..., _REST, :]           # [...,24,6,2]
    x1, y1 = ln[..., 0:1, 0], ln[..., 0:1, 1]
    x2, y2 = ln[..., 1:2, 0], ln[..., 1:2, 1]
    # projection range along the edge (:117-134)
    k = (y2 - y1) / (x2 - x1 + 1e-8)
    vec = torch.cat([torch.ones_like(k), k], dim=-1).unsqueeze(-2)
    allp = torch.cat([ln, pt], dim=-2)
    proj = (allp * vec).sum(-1) / torch.norm(vec, dim=-1)
    prange = proj.max(-1)[0] - proj.min(-1)[0]
    # distance range perpendicular to it (:87-115)
    x, y = pt[..., 0], pt[..., 1]
    den = (y2 - y1) * x - (x2 - x1) * y + x2 * y1 - y2 * x1
    d = den / torch.sqrt((y2 - y1).square() + (x2 - x1).square() + 1e-14)
    drange = torch.max(d.max(-1)[0] - d.min(-1)[0], d.abs().max(-1)[0])
    area = prange * drange
    area = area + (area == 0).to(c8.dtype) * 1e8
    idx = area.min(dim=-1, keepdim=True)[1]
    return prange.gather(-1, idx).squeeze(-1).float(), drange.gather(-1, idx).squeeze(-1).float()


def giou_3d(b1, b2):
    """cal_giou_3d, oriented_iou_loss.py:109-126 -> (loss, giou, iou)."""
    iou, c1, c2, zr, u3 = iou_3d(b1, b2, verbose=True)
    w, h = enclosing_box_wh(torch.cat([c1, c2], dim=-2))
    vc = zr * w * h
    loss = 1.0 - iou + (vc - u3) / vc
    return loss, 1 - loss, iou


def diou_3d(b1, b2):
    """cal_diou_3d, oriented_iou_loss.py:128-148 -> (loss, iou)."""
    iou, c1, c2, zr, u3 = iou_3d(b1, b2, verbose=True)
    w, h = enclosing_box_wh(torch.cat([c1, c2], dim=-2))
    dx, dy, dz = b1[..., 0] - b2[..., 0], b1[..., 1] - b2[..., 1], b1[..., 2] - b2[..., 2]
    d2 = dx * dx + dy * dy + dz * dz
    c2_ = w * w + h * h + zr * zr
    return 1.0 - iou + d2 / c2_, iou

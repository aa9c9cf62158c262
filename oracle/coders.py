"""ORACLE (test infrastructure only): anchor<->box coders on CPU.

Restates reference nerf_rpn/model/coder/AABB_coder.py:7-137 (AABB),
midpoint_offset_coder.py:106-222 + misc.py:3-93 (OBB as AABB + two midpoint offsets).
"""
import math

import numpy as np
import torch

PI_REF = 3.141592  # misc.py:3 -- the truncated constant is part of the behaviour (quirk B11)


# ----------------------------------------------------------------------------- AABB
def aabb_encode(gt, anchors):
    """AABB_coder.py:7-56: both [M,6] -> deltas [M,6]."""
    aw = anchors[:, 3:6] - anchors[:, 0:3]
    ac = anchors[:, 0:3] + 0.5 * aw
    gw = gt[:, 3:6] - gt[:, 0:3]
    gc = gt[:, 0:3] + 0.5 * gw
    return torch.cat([(gc - ac) / aw, torch.log(gw / aw)], dim=1)


def aabb_decode(deltas, anchors, clip=math.log(2000.0)):
    """AABB_coder.py:86-137: [M,6],[M,6] -> [M,6]."""
    anchors = anchors.to(deltas.dtype)
    aw = anchors[:, 3:6] - anchors[:, 0:3]
    ac = anchors[:, 0:3] + 0.5 * aw
    d = deltas[:, 3:6].clamp(max=clip)
    c = deltas[:, 0:3] * aw + ac
    half = 0.5 * (torch.exp(d) * aw)
    return torch.cat([c - half, c + half], dim=1)


# ----------------------------------------------------------------------------- OBB helpers
def obb2d_to_hbb(o):
    """misc.py:74-83: [.,5] (x,y,w,h,t) -> [.,4]."""
    c, w, h, t = o[..., 0:2], o[..., 2:3], o[..., 3:4], o[..., 4:5]
    cs, sn = torch.cos(t), torch.sin(t)
    bias = torch.cat([torch.abs(w / 2 * cs) + torch.abs(h / 2 * sn),
                      torch.abs(w / 2 * sn) + torch.abs(h / 2 * cs)], dim=-1)
    return torch.cat([c - bias, c + bias], dim=-1)


def obb2d_to_poly(o):
    """misc.py:50-62."""
    c, w, h, t = o[..., 0:2], o[..., 2:3], o[..., 3:4], o[..., 4:5]
    cs, sn = torch.cos(t), torch.sin(t)
    v1 = torch.cat([w / 2 * cs, -w / 2 * sn], dim=-1)
    v2 = torch.cat([-h / 2 * sn, -h / 2 * cs], dim=-1)
    return torch.cat([c + v1 + v2, c + v1 - v2, c - v1 - v2, c - v1 + v2], dim=-1)


def obb3d_to_hbb(o):
    """obb2hbb_3d, misc.py:85-93: [.,7] -> [.,6]."""
    c, z, w, h, d, t = o[..., 0:2], o[..., 2:3], o[..., 3:4], o[..., 4:5], o[..., 5:6], o[..., 6:7]
    cs, sn = torch.cos(t), torch.sin(t)
    bias = torch.cat([torch.abs(w / 2 * cs) + torch.abs(h / 2 * sn),
                      torch.abs(w / 2 * sn) + torch.abs(h / 2 * cs)], dim=-1)
    return torch.cat([c - bias, z - d / 2, c + bias, z + d / 2], dim=-1)


def obb3d_extreme_points(o):
    """obb2points_3d, misc.py:95-101: [K,7] -> [2K,3]."""
    c, w, l, h, t = o[..., 0:3], o[..., 3:4], o[..., 4:5], o[..., 5:6], o[..., 6:7]
    cs, sn = torch.cos(t), torch.sin(t)
    v = torch.cat([w / 2 * cs - l / 2 * sn, w / 2 * sn + l / 2 * cs, h / 2], dim=-1)
    return torch.cat([c - v, c + v], dim=0)


def _rectpoly_to_obb(p):
    """rectpoly2obb + regular_obb, misc.py:5-47.  p: [.,8] -> [.,5]."""
    t = torch.atan2(-(p[..., 3] - p[..., 1]), p[..., 2] - p[..., 0] + 1e-7)
    cs, sn = torch.cos(t), torch.sin(t)
    x = p[..., 0::2].mean(-1)
    y = p[..., 1::2].mean(-1)
    px = p[..., 0::2] - x.unsqueeze(-1)
    py = p[..., 1::2] - y.unsqueeze(-1)
    rx = px * cs.unsqueeze(-1) + py * (-sn).unsqueeze(-1)
    ry = px * sn.unsqueeze(-1) + py * cs.unsqueeze(-1)
    w = rx.max(-1)[0] - rx.min(-1)[0]
    h = ry.max(-1)[0] - ry.min(-1)[0]
    wr = torch.where(w > h, w, h)
    hr = torch.where(w > h, h, w)
    tr = torch.where(w > h, t, t + PI_REF / 2)
    tr = (tr + PI_REF / 2) % PI_REF - PI_REF / 2
    return torch.stack([x, y, wr, hr, tr], dim=-1)


# ----------------------------------------------------------------------------- midpoint offset
def midpoint_encode(gt, anchors):
    """bbox2delta_sp, midpoint_offset_coder.py:106-158: gt [M,7], anchors [M,6] -> [M,8]."""
    p, g = anchors.float(), gt.float()
    pc = (p[:, 0:3] + p[:, 3:6]) * 0.5
    pw = p[:, 3:6] - p[:, 0:3]
    g2 = torch.cat([g[:, 0:2], g[:, 3:5], g[:, 6:7]], dim=-1)
    hbb, poly = obb2d_to_hbb(g2), obb2d_to_poly(g2)
    gx, gy = (hbb[:, 0:1] + hbb[:, 2:3]) * 0.5, (hbb[:, 1:2] + hbb[:, 3:4]) * 0.5
    gw, gh = hbb[:, 2:3] - hbb[:, 0:1], hbb[:, 3:4] - hbb[:, 1:2]
    xs, ys = poly[:, 0::2], poly[:, 1::2]
    ymin = ys.min(dim=1, keepdim=True)[0]
    xmax = xs.max(dim=1, keepdim=True)[0]
    ga = torch.where((ys - ymin).abs() > 0.1, torch.full_like(xs, -1000.0), xs).max(dim=1, keepdim=True)[0]
    gb = torch.where((xs - xmax).abs() > 0.1, torch.full_like(ys, -1000.0), ys).max(dim=1, keepdim=True)[0]
    return torch.cat([(gx - pc[:, 0:1]) / pw[:, 0:1], (gy - pc[:, 1:2]) / pw[:, 1:2],
                      (g[:, 2:3] - pc[:, 2:3]) / pw[:, 2:3],
                      torch.log(gw / pw[:, 0:1]), torch.log(gh / pw[:, 1:2]), torch.log(g[:, 5:6] / pw[:, 2:3]),
                      (ga - gx) / gw, (gb - gy) / gh], dim=-1)


def midpoint_decode(deltas, anchors, ratio_clip=16 / 1000):
    """delta_sp2bbox, midpoint_offset_coder.py:160-222: [M,8],[M,6] -> [M,7]."""
    lim = float(np.abs(np.log(ratio_clip)))
    dxyz = deltas[:, 0:3]
    dwhd = deltas[:, 3:6].clamp(min=-lim, max=lim)
    da = deltas[:, 6:7].clamp(min=-0.5, max=0.5)
    db = deltas[:, 7:8].clamp(min=-0.5, max=0.5)
    pc = (anchors[:, 0:3] + anchors[:, 3:6]) * 0.5
    pw = anchors[:, 3:6] - anchors[:, 0:3]
    gs = pw * dwhd.exp()
    gc = pc + pw * dxyz
    gx, gy, gz = gc[:, 0:1], gc[:, 1:2], gc[:, 2:3]
    gw, gh, gd = gs[:, 0:1], gs[:, 1:2], gs[:, 2:3]
    x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
    ga, ga_ = gx + da * gw, gx - da * gw
    gb, gb_ = gy + db * gh, gy - db * gh
    poly = torch.cat([ga, y1, x2, gb, ga_, y2, x1, gb_], dim=-1)
    ctr = torch.cat([gx, gy] * 4, dim=-1)
    cp = poly - ctr
    diag = torch.sqrt(cp[:, 0::2] ** 2 + cp[:, 1::2] ** 2)
    scale = diag.max(dim=-1, keepdim=True)[0] / diag
    rect = cp * scale.repeat_interleave(2, dim=-1) + ctr
    o2 = _rectpoly_to_obb(rect)
    return torch.cat([o2[:, 0:2], gz, o2[:, 2:4], gd, o2[:, 4:5]], dim=-1)

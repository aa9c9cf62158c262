"""FCOS box utilities with the reference's names (reference nerf_rpn/model/fcos/utils.py).

``decode_fcos_obb`` is the differentiable torch formulation used by the rotated IoU losses on the positive locations (the
inference path decodes inside ``nrpn_fcos_decode_f32``); ``encode_fcos_obb`` runs the per-GT part in
``nrpn_fcos_gt_summary_f32``.  NMS / IoU / clipping are the shared HIP implementations of ``model/utils.py``."""
import torch

from ... import ops
from ..utils import nms, batched_nms, remove_small_boxes, clip_boxes_to_mesh, batched_box_iou, box_iou_3d  # noqa: F401
from ..rpn import _view_matrices


def decode_fcos_obb(locations, box_regression):
    """locations [N,3], box_regression [N,8] (l,t,f,r,b,ba,alpha,beta) -> OBB [N,7]; reference fcos/utils.py:12-62."""
    assert box_regression.shape[1] == 8, "box_regression for OBB should have 8 offsets"
    r = box_regression
    x0, y0, z0 = locations[:, 0] - r[:, 0], locations[:, 1] - r[:, 1], locations[:, 2] - r[:, 2]
    x1, y1, z1 = locations[:, 0] + r[:, 3], locations[:, 1] + r[:, 4], locations[:, 2] + r[:, 5]
    vx = torch.clamp((x1 + x0) / 2 + r[:, 6] * (x1 - x0), min=x0, max=x1)
    vy = torch.clamp((y1 + y0) / 2 + r[:, 7] * (y1 - y0), min=y0, max=y1)
    ctr = torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (z0 + z1) / 2], dim=1)
    v0 = torch.stack([vx, y1], dim=1) - ctr[:, :2]
    v1 = torch.stack([x1, vy], dim=1) - ctr[:, :2]
    d0, d1 = torch.norm(v0, dim=1), torch.norm(v1, dim=1)
    dmax = torch.max(d0, d1)
    v0 = v0 / (d0[:, None] + 1e-7) * dmax[:, None] + ctr[:, :2]
    v1 = v1 / (d1[:, None] + 1e-7) * dmax[:, None] + ctr[:, :2]
    length = torch.norm(v0 - v1, dim=1)
    mid = (v0 + v1) / 2 - ctr[:, :2]
    width = torch.norm(mid, dim=1) * 2
    zero = (mid[:, 0] == 0) & (mid[:, 1] == 0)
    mx = torch.where(zero, torch.full_like(mid[:, 0], 1e-7), mid[:, 0])
    return torch.stack([ctr[:, 0], ctr[:, 1], ctr[:, 2], width, length, z1 - z0, torch.atan2(mid[:, 1], mx)], dim=1)


def encode_fcos_obb(locations, boxes):
    """reference fcos/utils.py:65-105: row i pairs location i with box i."""
    assert boxes.shape[1] == 7, "input OBB should have 7 parameters"
    assert boxes.shape[0] == locations.shape[0], "number of boxes should be equal to number of locations"
    s = ops.fcos_gt_summary(boxes)
    return torch.cat([locations - s[:, :3], s[:, 3:6] - locations, s[:, 6:8]], dim=1)


def get_w2cs(res: int = 160, device="cuda"):
    return _view_matrices(res, torch.device(device))


def obb2points_3d(obboxes):
    center, w, l, h, theta = torch.split(obboxes, [3, 1, 1, 1, 1], dim=-1)
    c, s = torch.cos(theta), torch.sin(theta)
    vec = torch.cat([w / 2 * c - l / 2 * s, w / 2 * s + l / 2 * c, h / 2], dim=-1)
    return torch.cat([center - vec, center + vec], dim=0)


def project(intrinsic_mat, pose_mat, box_coords):
    cam = torch.matmul(pose_mat, torch.transpose(box_coords, 0, 1).float())
    pic = torch.matmul(intrinsic_mat, cam[:3, :])
    return torch.transpose(pic[:2, :] / pic[2, :], 0, 1)

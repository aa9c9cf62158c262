"""Second-stage box coder for z-rotated boxes (reference nerf_rpn/model/coder/rotated_coder.py:13-128): deltas are expressed in the
RoI's own rotated frame, sizes as log ratios (clamped at log 2000 when decoding), the angle as a fraction of 2*pi.  A few hundred
RoIs per step: plain torch ops on the device."""
import math

import torch
from torch import Tensor

from .coders import BaseBBoxCoder


class RotatedCoder(BaseBBoxCoder):
    def __init__(self, bbox_xform_clip: float = math.log(2000.0)):
        self.bbox_xform_clip = bbox_xform_clip

    def encode_single(self, gt_rois: Tensor, ex_drois: Tensor) -> Tensor:
        """gt, rois [N,7] (x, y, z, w, h, d, theta) -> [N,7] (dx, dy, dz, dw, dh, dd, dtheta)."""
        d = gt_rois[:, 0:3] - ex_drois[:, 0:3]
        c, s = torch.cos(ex_drois[:, 6]), torch.sin(ex_drois[:, 6])
        dx = (c * d[:, 0] + s * d[:, 1]) / ex_drois[:, 3]
        dy = (-s * d[:, 0] + c * d[:, 1]) / ex_drois[:, 4]
        dz = d[:, 2] / ex_drois[:, 5]
        dw = torch.log(gt_rois[:, 3] / ex_drois[:, 3])
        dh = torch.log(gt_rois[:, 4] / ex_drois[:, 4])
        dd = torch.log(gt_rois[:, 5] / ex_drois[:, 5])
        da = (gt_rois[:, 6] - ex_drois[:, 6]) / (2 * torch.pi)
        return torch.stack((dx, dy, dz, dw, dh, dd, da), 1)

    def decode_single(self, deltas: Tensor, ex_drois: Tensor) -> Tensor:
        """deltas [N, 7k], rois [N,7] -> boxes [N, 7k]; theta wrapped to (-pi/2, pi/2]."""
        assert deltas.size(0) == ex_drois.size(0)
        clip = torch.tensor(self.bbox_xform_clip, device=deltas.device)
        cx, cy, cz, w, h, dpt, ang = [ex_drois[:, i, None] for i in range(7)]
        dx, dy, dz, dw, dh, dd, da = [deltas[:, i::7] for i in range(7)]
        dw, dh, dd = torch.min(dw, clip), torch.min(dh, clip), torch.min(dd, clip)
        px = dx * w * torch.cos(ang) - dy * h * torch.sin(ang) + cx
        py = dx * w * torch.sin(ang) + dy * h * torch.cos(ang) + cy
        pz = dz * dpt + cz
        pa = ((2 * torch.pi) * da + ang) % torch.pi
        pa = torch.where(pa > torch.pi / 2, pa - torch.pi, pa)
        out = torch.ones_like(deltas)
        out[:, 0::7], out[:, 1::7], out[:, 2::7] = px, py, pz
        out[:, 3::7], out[:, 4::7], out[:, 5::7] = torch.exp(dw) * w, torch.exp(dh) * h, torch.exp(dd) * dpt
        out[:, 6::7] = pa
        return out

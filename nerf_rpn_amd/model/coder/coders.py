"""Box coders with the reference's class API (coder/base_bbox_coder.py, AABB_coder.py, midpoint_offset_coder.py).

``encode_single`` / ``decode_single`` run the HIP coder kernels (no autograd).  ``decode_single_diff`` is a torch
formulation used only where a gradient through the decode is required (IoU-type regression losses, 2-D projection
loss) on the <= 128 sampled rows per scene."""
import math

import numpy as np
import torch
from torch import Tensor

from ... import ops
from .misc import pi


class BaseBBoxCoder:
    coder_id = 0

    def encode_single(self, bboxes: Tensor, proposals: Tensor) -> Tensor:
        return ops.coder_pairs(bboxes, proposals, self.coder_id, True)

    def decode_single(self, deltas: Tensor, proposals: Tensor) -> Tensor:
        return ops.coder_pairs(deltas.detach(), proposals, self.coder_id, False)

    def encode(self, bboxes_ref, proposals):
        per = [len(b) for b in bboxes_ref]
        out = self.encode_single(torch.cat(bboxes_ref, dim=0), torch.cat(proposals, dim=0))
        return out.split(per, 0)

    def decode(self, bboxes_ref: Tensor, deltas):
        total = sum(b.size(0) for b in deltas)
        anchors = torch.cat(deltas, dim=0)
        if total > 0:
            bboxes_ref = bboxes_ref.reshape(total, -1)
        pred = self.decode_single(bboxes_ref, anchors)
        return pred[:, None, :] if total > 0 else pred

    def decode_list(self, delta_list, boxes_list, is_cat=True):
        out = []
        for i, rel in enumerate(delta_list):
            cur = []
            for j in range(rel.size(0)):
                boxes = boxes_list[j][i] if isinstance(boxes_list[j], list) else boxes_list[j]
                pred = self.decode_single(rel[j].detach(), boxes.detach())
                cur.append(torch.cat([pred, pred.new_full((pred.size(0), 1), float(i))], dim=1))
            out.append(torch.stack(cur))
        return torch.cat(out, dim=1) if is_cat else out


class AABBCoder(BaseBBoxCoder):
    coder_id = 0

    def __init__(self, bbox_xform_clip: float = math.log(2000.0)) -> None:
        if abs(bbox_xform_clip - math.log(2000.0)) > 1e-12:
            raise NotImplementedError("the HIP AABB coder has the reference clip log(2000) built in")
        self.bbox_xform_clip = bbox_xform_clip

    def decode_single_diff(self, d: Tensor, anchors: Tensor) -> Tensor:
        aw = anchors[:, 3:6] - anchors[:, 0:3]
        ac = anchors[:, 0:3] + 0.5 * aw
        c = d[:, 0:3] * aw + ac
        half = 0.5 * (torch.exp(d[:, 3:6].clamp(max=self.bbox_xform_clip)) * aw)
        return torch.cat([c - half, c + half], dim=1)


class MidpointOffsetCoder(BaseBBoxCoder):
    coder_id = 1

    def __init__(self, target_means=(0.,) * 8, target_stds=(1.,) * 8):
        if any(m != 0 for m in target_means) or any(s != 1 for s in target_stds):
            raise NotImplementedError("non-default target means/stds are never used by the reference RPN")
        self.means, self.stds = target_means, target_stds

    def decode_single_diff(self, deltas: Tensor, anchors: Tensor, wh_ratio_clip=16 / 1000) -> Tensor:
        lim = float(np.abs(np.log(wh_ratio_clip)))
        dwhd = deltas[:, 3:6].clamp(min=-lim, max=lim)
        da = deltas[:, 6:7].clamp(min=-0.5, max=0.5)
        db = deltas[:, 7:8].clamp(min=-0.5, max=0.5)
        pc = (anchors[:, 0:3] + anchors[:, 3:6]) * 0.5
        pw = anchors[:, 3:6] - anchors[:, 0:3]
        gs = pw * dwhd.exp()
        gc = pc + pw * deltas[:, 0:3]
        gx, gy, gz = gc[:, 0:1], gc[:, 1:2], gc[:, 2:3]
        gw, gh, gd = gs[:, 0:1], gs[:, 1:2], gs[:, 2:3]
        x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
        poly = torch.cat([gx + da * gw, y1, x2, gy + db * gh, gx - da * gw, y2, x1, gy - db * gh], dim=-1)
        ctr = torch.cat([gx, gy] * 4, dim=-1)
        cp = poly - ctr
        diag = torch.sqrt(cp[:, 0::2] ** 2 + cp[:, 1::2] ** 2)
        rect = cp * (diag.max(dim=-1, keepdim=True)[0] / diag).repeat_interleave(2, dim=-1) + ctr
        t = torch.atan2(-(rect[:, 3] - rect[:, 1]), rect[:, 2] - rect[:, 0] + 1e-7)
        cs, sn = torch.cos(t), torch.sin(t)
        x, y = rect[:, 0::2].mean(-1), rect[:, 1::2].mean(-1)
        px, py = rect[:, 0::2] - x[:, None], rect[:, 1::2] - y[:, None]
        rx = px * cs[:, None] - py * sn[:, None]
        ry = px * sn[:, None] + py * cs[:, None]
        w = rx.max(-1)[0] - rx.min(-1)[0]
        h = ry.max(-1)[0] - ry.min(-1)[0]
        wide = w > h
        tr = torch.where(wide, t, t + pi / 2)
        tr = (tr + pi / 2) % pi - pi / 2
        o2 = torch.stack([x, y, torch.where(wide, w, h), torch.where(wide, h, w), tr], dim=-1)
        return torch.cat([o2[:, 0:2], gz, o2[:, 2:4], gd, o2[:, 4:5]], dim=-1)

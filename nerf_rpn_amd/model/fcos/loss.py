"""FCOS losses on the HIP kernels (reference nerf_rpn/model/fcos/loss.py:77-591).

Targets for every location (centre sampling, size-of-interest ranges, smallest-volume GT) and the focal loss with its
gradient are single kernels over the ~70 k locations; the regression / centerness terms touch only the few hundred positive
locations and stay in torch ops (the rotated IoU variants use the differentiable polygon clipping of
``rotated_iou/oriented_iou_loss.py`` with the HIP vertex sort, as the RPN losses do)."""
import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ..rotated_iou.oriented_iou_loss import cal_iou_3d, cal_giou_3d, cal_diou_3d
from .utils import decode_fcos_obb, get_w2cs, obb2points_3d, project

INF = 100000000


class IOULoss(nn.Module):
    """3D IoU / linear IoU / GIoU loss on (l, t, f, r, b, ba) distances; reference loss.py:77-134."""

    def __init__(self, loss_type="iou"):
        super().__init__()
        self.loss_type = loss_type

    def forward(self, pred, target, weight=None):
        pl, pt, pf, pr, pb, pk = pred.unbind(1)
        tl, tt, tf, tr, tb, tk = target.unbind(1)
        target_volume = (tl + tr) * (tt + tb) * (tf + tk)
        pred_volume = (pl + pr) * (pt + pb) * (pf + pk)
        w_i = torch.min(pl, tl) + torch.min(pr, tr)
        g_w = torch.max(pl, tl) + torch.max(pr, tr)
        h_i = torch.min(pb, tb) + torch.min(pt, tt)
        g_h = torch.max(pb, tb) + torch.max(pt, tt)
        d_i = torch.min(pf, tf) + torch.min(pk, tk)
        g_d = torch.max(pf, tf) + torch.max(pk, tk)
        ac_union = g_w * g_h * g_d + 1e-7
        volume_intersect = w_i * h_i * d_i
        volume_union = target_volume + pred_volume - volume_intersect
        ious = (volume_intersect + 1.0) / (volume_union + 1.0)
        gious = ious - (ac_union - volume_union) / ac_union
        if self.loss_type == "iou":
            losses = -torch.log(ious)
        elif self.loss_type == "linear_iou":
            losses = 1 - ious
        elif self.loss_type == "giou":
            losses = 1 - gious
        else:
            raise NotImplementedError
        if weight is not None and weight.sum() > 0:
            return (losses * weight).sum()
        assert losses.numel() != 0
        return losses.sum()


class RotatedIOULoss(nn.Module):
    """reference loss.py:137-173."""

    def __init__(self, loss_type="iou"):
        super().__init__()
        self.loss_type = loss_type

    def forward(self, pred, target, weight=None):
        dummy = torch.zeros(pred.shape[0], 3, device=pred.device)
        pb, tb = decode_fcos_obb(dummy, pred).unsqueeze(0), decode_fcos_obb(dummy, target).unsqueeze(0)
        if pb.is_cuda and not tb.requires_grad and self.loss_type in ("iou", "linear_iou", "giou", "diou"):
            losses = ops.rotated_iou_loss(pb[0], tb[0], self.loss_type)[0]        # fused forward + gradient (csrc/geomloss.hip)
            if weight is not None and weight.sum() > 0:
                return (losses * weight).sum()
            assert losses.numel() != 0
            return losses.sum()
        if self.loss_type in ("iou", "linear_iou"):
            ious, _, _, _, unions = cal_iou_3d(pb, tb, verbose=True)
            ious = (ious * unions + 1.0) / (unions + 1.0)
            losses = -torch.log(ious) if self.loss_type == "iou" else 1 - ious
        elif self.loss_type == "giou":
            losses, _, _ = cal_giou_3d(pb, tb)
        elif self.loss_type == "diou":
            losses, _ = cal_diou_3d(pb, tb)
        else:
            raise NotImplementedError
        if weight is not None and weight.sum() > 0:
            return (losses * weight).sum()
        assert losses.numel() != 0
        return losses.sum()


class FCOSLossComputation(object):
    def __init__(self, fpn_strides, center_sampling_radius, iou_loss_type, norm_reg_targets, world_size, use_obb, use_additional_l1_loss,
                 proj2d_loss_weight=0.0):
        self.fpn_strides, self.center_sampling_radius, self.iou_loss_type = fpn_strides, center_sampling_radius, iou_loss_type
        self.norm_reg_targets, self.world_size, self.use_obb = norm_reg_targets, world_size, use_obb
        self.use_additional_l1_loss, self.proj2d_loss_weight = use_additional_l1_loss, proj2d_loss_weight
        if iou_loss_type != "smooth_l1":
            self.box_reg_loss_func = IOULoss(iou_loss_type) if not use_obb else RotatedIOULoss(iou_loss_type)
        else:
            self.box_reg_loss_func = nn.SmoothL1Loss(reduction="none")
        self.centerness_loss_func = nn.BCEWithLogitsLoss(reduction="sum")
        self.additional_l1_loss_func = nn.SmoothL1Loss(reduction="none")
        self.last_aux = None

    def reduce_sum(self, tensor):
        if self.world_size <= 1:
            return tensor
        import torch.distributed as dist
        tensor = tensor.clone()
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        return tensor

    def normalisers(self, num_pos, ctr_sum):
        """(positives per rank averaged over the ranks, clamped at 1; centerness-target sum averaged over the ranks) -- the two scalars the
        data-parallel FCOS loss exchanges per step (reference fcos/loss.py:533-550).  ``ctr_sum`` None = this rank has no positive location:
        it still takes part in the second reduction (the reference's dummy reduce, loss.py:588), otherwise the other ranks would hang."""
        n = float(self.world_size)
        num_pos_avg = max(self.reduce_sum(num_pos.to(torch.int64).reshape(1)).item() / n, 1.0)
        if ctr_sum is None:
            ctr_sum = torch.zeros((), dtype=torch.float32, device=num_pos.device)
        norm = self.reduce_sum(ctr_sum.detach().float().reshape(1)).item() / n
        return num_pos_avg, norm

    def prepare_targets(self, geom, targets, pad_sizes, device):
        """labels int8 [total] {1, 0, -1 = padding}, reg_targets [total, 6|8] (stride-normalised), num_pos int32 [1]."""
        return ops.fcos_targets(geom, [t.float() for t in targets], pad_sizes, self.center_sampling_radius, self.norm_reg_targets,
                                8 if self.use_obb else 6, device)

    @staticmethod
    def compute_centerness_targets(reg_targets):
        lr, tb, fb = reg_targets[:, [0, 3]], reg_targets[:, [1, 4]], reg_targets[:, [2, 5]]
        c = (lr.min(dim=-1)[0] / lr.max(dim=-1)[0]) * (tb.min(dim=-1)[0] / tb.max(dim=-1)[0]) * (fb.min(dim=-1)[0] / fb.max(dim=-1)[0])
        return torch.sqrt(c)

    def compute_2d_projection_loss(self, box_reg, reg_targets, weights):
        dev = box_reg.device
        K = torch.tensor([[600., 0., 320.], [0., 600., 240.], [0., 0., 1.]], device=dev)
        dummy = torch.zeros(box_reg.shape[0], 3, device=dev)
        p, t = obb2points_3d(decode_fcos_obb(dummy, box_reg)), obb2points_3d(decode_fcos_obb(dummy, reg_targets))
        one = torch.ones(p.shape[0], 1, device=dev)
        p, t = torch.cat([p, one], dim=1), torch.cat([t, one], dim=1)
        poses = get_w2cs(160, dev)
        p2, t2 = torch.cat([project(K, M, p) for M in poses]), torch.cat([project(K, M, t) for M in poses])
        loss = F.smooth_l1_loss(p2, t2, beta=1 / 9, reduction="none") / 160
        factor = loss.shape[0] // weights.shape[0]
        return (loss * weights[:, None].repeat(factor, 1)).sum() / (factor * loss.shape[1])

    def prepare(self, geom, targets, pad_sizes, device):
        """Everything of the loss that does not depend on the network's outputs -- target assignment, the positive-location list, the centerness
        targets and the two normalisers -- and with it EVERY host synchronisation of an FCOS training step (``nonzero`` + two ``.item()``).
        FCOSOverNeRF runs it BEFORE the backbone on its own stream (round 5): the forward, the rest of the loss and the backward are then
        enqueued without a synchronisation, instead of the host waiting for the head's outputs in the middle of the step."""
        labels, reg_targets, num_pos = self.prepare_targets(geom, targets, pad_sizes, device)
        pos_inds = torch.nonzero(labels > 0).squeeze(1)
        rt_p = reg_targets[pos_inds]
        ctr_t = self.compute_centerness_targets(rt_p) if pos_inds.numel() else None
        num_pos_avg, norm = self.normalisers(num_pos, ctr_t.sum() if ctr_t is not None else None)
        return {"labels": labels, "reg_targets": reg_targets, "pos": pos_inds, "rt_p": rt_p, "ctr_t": ctr_t, "num_pos_avg": num_pos_avg, "norm": norm}

    def __call__(self, geom, logits, reg, ctr, targets, pad_sizes, prepared=None):
        """logits / ctr [total], reg [total, 6|8] flattened (level, scene, voxel) -> (cls_loss, reg_loss, centerness_loss)."""
        pr = prepared if prepared is not None else self.prepare(geom, targets, pad_sizes, logits.device)
        labels, reg_targets, pos_inds, rt_p, ctr_t = pr["labels"], pr["reg_targets"], pr["pos"], pr["rt_p"], pr["ctr_t"]
        num_pos_avg, norm = pr["num_pos_avg"], pr["norm"]
        reg_p, ctr_p = reg[pos_inds], ctr[pos_inds]
        cls_loss = ops.FocalLossFn.apply(logits, labels, 0.25) / num_pos_avg
        self.last_aux = {"labels": labels, "reg_targets": reg_targets, "pos": pos_inds}
        if pos_inds.numel() == 0:
            return cls_loss, reg_p.sum(), ctr_p.sum()
        if self.iou_loss_type != "smooth_l1":
            reg_loss = self.box_reg_loss_func(reg_p, rt_p, ctr_t) / norm
        else:
            reg_loss = (self.box_reg_loss_func(reg_p, rt_p) * ctr_t.unsqueeze(1)).sum() / norm
        ctr_loss = self.centerness_loss_func(ctr_p, ctr_t) / num_pos_avg
        if self.use_obb and self.use_additional_l1_loss and self.iou_loss_type != "smooth_l1":
            reg_loss = reg_loss + (self.additional_l1_loss_func(reg_p[:, 6:], rt_p[:, 6:]) * ctr_t.unsqueeze(-1)).sum() / norm
        if self.use_obb and self.proj2d_loss_weight > 0:
            reg_loss = reg_loss + self.compute_2d_projection_loss(reg_p, rt_p, ctr_t) / norm * self.proj2d_loss_weight
        return cls_loss, reg_loss, ctr_loss

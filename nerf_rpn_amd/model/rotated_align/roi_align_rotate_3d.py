"""Rotated 3D RoIAlign with the reference's names and call contract (nerf_rpn/model/rotated_align/roi_align_rotate_3d.py:13-77):
``roi_align_rotated_3d(input [N,C,W,L,H], rois [R,8], output_size, spatial_scale, sampling_ratio) -> [R,C,pw,pl,ph]`` with
rois rows (batch index, cx, cy, cz, w, l, h, theta in degrees).  The arithmetic is the HIP kernel pair of csrc/roialign.hip on the
channels-last memory the backbones already produce (a channels-last-backed input is consumed as is, otherwise it is converted once);
the result is returned as a channels-last-backed view of the reference's logical shape."""
import torch
from torch import nn

from ... import ops


class _ROIAlignRotated3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        x = ops.as_channels_last(input)
        rois = roi.detach().float().contiguous()
        out = ops.roi_align_rotated_3d_fwd(x, rois, float(spatial_scale), tuple(int(v) for v in output_size), int(sampling_ratio))
        ctx.save_for_backward(rois)
        ctx.meta = (tuple(x.shape), tuple(int(v) for v in output_size), float(spatial_scale), int(sampling_ratio))
        return out.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        shape, output_size, spatial_scale, sampling_ratio = ctx.meta
        g = grad_output.permute(0, 2, 3, 4, 1).contiguous()
        gi = ops.roi_align_rotated_3d_bwd(g, rois, shape, spatial_scale, output_size, sampling_ratio)
        return gi.permute(0, 4, 1, 2, 3), None, None, None, None


roi_align_rotated_3d = _ROIAlignRotated3D.apply


class ROIAlignRotated3D(nn.Module):
    def __init__(self, output_size, sampling_ratio):
        super().__init__()
        self.output_size = output_size
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois, spatial_scale):
        return roi_align_rotated_3d(input, rois, self.output_size, spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return f"{self.__class__.__name__}(output_size={self.output_size}, sampling_ratio={self.sampling_ratio})"

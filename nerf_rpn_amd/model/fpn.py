"""Feature pyramid neck on the HIP kernels (reference nerf_rpn/model/fpn.py:8-185, default arguments only:
1x1x1 laterals, nearest top-down add, 3x3x3 output convs, no extra levels)."""
from torch import nn

from .. import ops
from . import hip_nn


class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 extra_convs_on_inputs=True, relu_before_extra_convs=False, upsample_cfg=dict(mode='nearest')):
        super().__init__()
        assert isinstance(in_channels, list)
        if start_level != 0 or end_level != -1 or add_extra_convs or num_outs != len(in_channels) \
                or upsample_cfg.get('mode', 'nearest') != 'nearest' or 'scale_factor' in upsample_cfg:
            raise NotImplementedError("HIP FPN implements the configuration the NeRF-RPN backbones use "
                                      "(all levels, no extra convs, nearest upsampling by size)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.lateral_convs = nn.ModuleList(nn.Conv3d(c, out_channels, 1) for c in in_channels)
        self.fpn_convs = nn.ModuleList(nn.Conv3d(out_channels, out_channels, 3, padding=1) for _ in in_channels)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward_cl(self, inputs):
        """inputs / outputs: channels-last tensors."""
        assert len(inputs) == len(self.in_channels)
        lat = [hip_nn.conv3d(c, x) for c, x in zip(self.lateral_convs, inputs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = ops.UpsampleAddFn.apply(lat[i - 1], lat[i])
        return tuple(hip_nn.conv3d(c, x) for c, x in zip(self.fpn_convs, lat))

    def forward(self, inputs):
        dt = inputs[0].dtype
        outs = self.forward_cl([hip_nn.as_ndhwc(x, dt) for x in inputs])
        return tuple(hip_nn.as_ncdhw(o) for o in outs)

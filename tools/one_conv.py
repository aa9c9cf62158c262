"""Run the dominant conv shape (256->256, 3^3, 40^3, bf16) a few times under every forward / dgrad kernel that can serve it -- the halo form
(the default for this shape), the 256x256 tile on 8 waves and on 4 waves (nrpn_conv_opts.tile) -- and the 256x256 wgrad kernel, for the
rocprofv3 --pmc passes of tools/pmc_conv.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops  # noqa: E402

dev = torch.device('cuda:0')
grid, cin, cout, k, dtype = 40, 256, 256, 3, torch.bfloat16
torch.manual_seed(0)
x = torch.randn(1, grid, grid, grid, cin, device=dev).clamp_min(0).to(dtype)          # post-ReLU activations, as in the network
dy = torch.randn(1, grid, grid, grid, cout, device=dev).to(dtype)
w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
wp, wpd = ops.PackedWeight().get([w], dtype, cout, True)
gw = torch.empty(lib.query('conv3d_wgrad_slices', 1, grid, grid, grid, cin, cout, cout, k, 1), 27, cout, cin, device=dev)
wsg = torch.empty(lib.query('conv3d_wgrad_workspace_bytes', 1, grid, grid, grid, cin, cout, cout, k, 1), dtype=torch.uint8, device=dev)
for _ in range(3):
    for tile in (lib.TILE_HALO, lib.TILE_256X256, lib.TILE_256X256_W4):
        ops._conv_fwd(x, wp, None, cout, cout, k, 0, dtype, tile=tile)
    lib.call('conv3d_wgrad', x.data_ptr(), dy.data_ptr(), gw.data_ptr(), 0, 1, grid, grid, grid, cin, cout, cout, k, 1, 0, wsg.data_ptr(), ops._s())
torch.cuda.synchronize()

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
# round 5: both K orders of the halo kernel (conv_halo_kernel<0, false, false> = 14 K-steps per chunk, <0, true, false> = taps paired across
# chunk boundaries: the default from Cin 256 up) and the bf16x3 form of the layer (Cin tripled, fp32 rows: <0, true, true>)
xf = torch.randn(1, grid, grid, grid, cin, device=dev).clamp_min(0)
xs = ops.split3(xf, 0b100)[0]
w3 = ops.split3(ops.PackedWeight().get([w], torch.float32, cout, False)[0], 0b010)[0]
for _ in range(3):
    for pairing in (2, 1):
        ops._conv_fwd(x, wp, None, cout, cout, k, 0, dtype, tile=lib.TILE_HALO, halo_pairing=pairing)
    ops._conv_fwd(xs, w3, None, cout, cout, k, 0, torch.float32, tile=lib.TILE_HALO)
    ops._conv_fwd(x, wp, None, cout, cout, k, 0, dtype, tile=lib.TILE_256X256)
    lib.call('conv3d_wgrad', x.data_ptr(), dy.data_ptr(), gw.data_ptr(), 0, 1, grid, grid, grid, cin, cout, cout, k, 1, 0, wsg.data_ptr(), ops._s())
torch.cuda.synchronize()

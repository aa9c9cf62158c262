"""Run the dominant conv shapes a few times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops
dev = torch.device('cuda:0')
grid, cin, cout, k, dtype = 40, 256, 256, 3, torch.bfloat16
x = torch.randn(1, grid, grid, grid, cin, device=dev).to(dtype)
dy = torch.randn(1, grid, grid, grid, cout, device=dev).to(dtype)
w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
wp, wpd = ops.PackedWeight().get([w], dtype, cout, True)
y = torch.empty_like(dy)
gw = torch.empty(lib.query('conv3d_wgrad_slices', 1, grid, grid, grid, cin, cout, cout, k, 1), 27, cout, cin, device=dev)
wsg = torch.empty(lib.query('conv3d_wgrad_workspace_bytes', 1, grid, grid, grid, cin, cout, cout, k, 1), dtype=torch.uint8, device=dev)
for _ in range(3):
    lib.call('conv3d_fwd', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), 1, grid, grid, grid, cin, cout, cout, k, 1, 0, 0, 0, ops._s())
    lib.call('conv3d_wgrad', x.data_ptr(), dy.data_ptr(), gw.data_ptr(), 0, 1, grid, grid, grid, cin, cout, cout, k, 1, 0, wsg.data_ptr(), ops._s())
torch.cuda.synchronize()

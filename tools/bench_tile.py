"""A/B of the 128- vs 256-row M tile of the bf16 LDS-DMA conv kernel on the heavy shapes (HIP events), with an equality check."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
dev = torch.device('cuda:0')
for grid, cin, cout, k in [(40, 256, 256, 3), (20, 512, 512, 3), (20, 256, 512, 3), (20, 256, 256, 3), (20, 512, 256, 1), (10, 512, 512, 3)]:
    x = torch.randn(1, grid, grid, grid, cin, device=dev).bfloat16()
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    wp, _ = ops.PackedWeight().get([w], torch.bfloat16, cout, False)
    flops = 2.0 * grid ** 3 * cin * cout * k ** 3
    outs = {}
    line = f'{grid}^3 {cin}->{cout} k{k}:'
    for bm in (128, 256, 0):
        lib.call('set_conv_tile_m', bm)
        wsb = lib.query('conv3d_fwd_workspace_bytes', 1, grid, grid, grid, cin, cout, k, ops._dt(x))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        y = torch.empty(1, grid, grid, grid, cout, device=dev, dtype=torch.bfloat16)
        fn = lambda: lib.call('conv3d_fwd', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), 1, grid, grid, grid, cin, cout, cout, k, ops._dt(x), 0, ws.data_ptr() if wsb else 0, 0, ops._s())
        t = timeit(fn, iters=30)
        outs[bm] = y.float()
        line += f'  bm{bm}(ws {wsb>>20}MB): {t*1e3:.0f} us {flops / t / 1e9:.0f} TF'
    print(line, ' maxdiff', (outs[128] - outs[256]).abs().max().item(), (outs[128] - outs[0]).abs().max().item())
lib.call('set_conv_tile_m', 0)

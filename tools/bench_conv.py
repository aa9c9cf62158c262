"""Micro-benchmark of the conv kernels on the dominant shapes (HIP events on the launch stream)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops
dev = torch.device('cuda:0')


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def run(grid, cin, cout, k, dtype):
    n = 1
    x = torch.randn(n, grid, grid, grid, cin, device=dev).to(dtype)
    dy = torch.randn(n, grid, grid, grid, cout, device=dev).to(dtype)
    w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    pk = ops.PackedWeight()
    wp, wpd = pk.get([w], dtype, cout, True)
    flops = 2.0 * n * grid ** 3 * cin * cout * k ** 3
    y = torch.empty(n, grid, grid, grid, cout, device=dev, dtype=dtype)
    gw = torch.empty(lib.query('conv3d_wgrad_slices', n, grid, grid, grid, cin, cout, cout, k, ops._dt(x)), k ** 3, cout, cin, device=dev)
    dt = ops._dt(x)
    wsb = lib.query('conv3d_fwd_workspace_bytes', n, grid, grid, grid, cin, cout, k, dt)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
    wgb = lib.query('conv3d_wgrad_workspace_bytes', n, grid, grid, grid, cin, cout, cout, k, dt)
    wsg = torch.empty(wgb, dtype=torch.uint8, device=dev) if wgb else None
    res = {}
    for kb, dma in ((128, 0), (64, 1), (128, 1)):
        lib.call('set_conv_kstep_bytes', kb)
        lib.call('set_conv_lds_dma', dma)
        t = timeit(lambda: lib.call('conv3d_fwd', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), n, grid, grid, grid, cin, cout, cout, k, dt, 0, ws.data_ptr() if ws is not None else 0, 0, ops._s()))
        res[f'fwd kb{kb}{"dma" if dma else "reg"}'] = (t, flops / t / 1e9)
    t = timeit(lambda: lib.call('conv3d_wgrad', x.data_ptr(), dy.data_ptr(), gw.data_ptr(), 0, n, grid, grid, grid, cin, cout, cout, k, dt, 0, wsg.data_ptr() if wsg is not None else 0, ops._s()))
    res['wgrad'] = (t, flops / t / 1e9)
    print(f'{grid}^3 {cin}->{cout} k{k} {str(dtype)[6:]}: ' + '  '.join(f'{a}: {v[0]*1e3:.0f} us {v[1]:.0f} TF' for a, v in res.items()))


for dtype in (torch.bfloat16, torch.float32):
    for shape in [(40, 256, 256, 3), (40, 128, 256, 3), (40, 64, 64, 3), (20, 512, 512, 3), (10, 512, 512, 3), (5, 256, 256, 3), (40, 256, 128, 1), (40, 128, 256, 1)]:
        run(*shape, dtype)

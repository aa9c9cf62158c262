import os, sys, torch
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
for dtype in (torch.bfloat16, torch.float32):
    for grid, cin, cout in [(40, 256, 256), (40, 128, 128), (40, 256, 128), (20, 256, 256)]:
        x = torch.randn(1, grid, grid, grid, cin, device=dev).to(dtype)
        dy = torch.randn(1, grid, grid, grid, cout, device=dev).to(dtype)
        S = lib.query('conv3d_wgrad_slices', 1, grid, grid, grid, cin, cout, cout, 1, ops._dt(x))
        gw = torch.empty(S, 1, cout, cin, device=dev)
        gb = torch.empty(cout, device=dev)
        line = f'{grid}^3 {cin}->{cout} k1 {str(dtype)[6:]} S={S}:'
        for tr in (1, 0):
            lib.call('set_wgrad_transpose_read', tr)
            for bias in (1, 0):
                t = timeit(lambda: lib.call('conv3d_wgrad', x.data_ptr(), dy.data_ptr(), gw.data_ptr(), gb.data_ptr() if bias else 0, 1, grid, grid, grid, cin, cout, cout, 1, ops._dt(x), 0, 0, ops._s()))
                line += f'  tr{tr} bias{bias}: {t*1e3:.0f} us'
        lib.call('set_wgrad_transpose_read', 1)
        print(line)

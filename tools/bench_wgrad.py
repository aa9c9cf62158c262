"""A/B of the 128x128 vs 256x256 wgrad tile (bf16) on the wide layers: time (HIP events) and agreement of dW / dbias."""
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


for grid, cin, cout, k in [(40, 256, 256, 3), (20, 512, 512, 3), (20, 256, 512, 3), (10, 512, 512, 3), (40, 256, 256, 1), (21, 256, 320, 3)]:
    g3 = grid ** 3
    x = torch.randn(1, grid, grid, grid, cin, device=dev).bfloat16()
    dy = (torch.randn(1, grid, grid, grid, cout, device=dev) * (torch.rand(1, grid, grid, grid, 1, device=dev) < 0.3)).bfloat16()
    flops = 2.0 * g3 * cin * cout * k ** 3
    wgb = lib.query('conv3d_wgrad_workspace_bytes', 1, grid, grid, grid, cin, cout, cout, k, 1)
    ws = torch.empty(max(wgb, 16), dtype=torch.uint8, device=dev)
    outs = {}
    line = f'{grid}^3 {cin}->{cout} k{k}:'
    for big in (0, 1):
        lib.call('set_wgrad_big_tile', big)
        S = lib.query('conv3d_wgrad_slices', 1, grid, grid, grid, cin, cout, cout, k, ops._dt(x))
        gw = torch.empty(S, k ** 3, cout, cin, device=dev)
        gb = torch.empty(cout, device=dev)
        fn = lambda: lib.call('conv3d_wgrad', x.data_ptr(), dy.data_ptr(), gw.data_ptr(), gb.data_ptr(), 1, grid, grid, grid, cin, cout, cout, k,
                              ops._dt(x), 0, ws.data_ptr(), ops._s())
        t = timeit(fn)
        outs[big] = (gw.sum(0), gb.clone())
        line += f' S={S}'
        line += f'  big{big}: {t*1e3:.0f} us {flops / t / 1e9:.0f} TF'
    dw = (outs[0][0] - outs[1][0]).abs().max().item() / outs[0][0].abs().max().item()
    db = (outs[0][1] - outs[1][1]).abs().max().item() / (outs[0][1].abs().max().item() + 1e-9)
    print(line, f' rel diff dW {dw:.1e} dbias {db:.1e}')
lib.call('set_wgrad_big_tile', 1)

"""Timing diagnosis of conv_igemm_big_kernel on the dominant shape (256->256, 3^3, 40^3, bf16): baseline vs "ideal memory" (every tap
reads the centre voxel) vs "free-running waves" (no per-K-step barrier / DMA drain) vs both.  Results of the debug modes are wrong
by construction; only the durations are meaningful.  Also the 128-row kernel at 64- vs 128-byte K-steps (half- vs full-line fetches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


dev = torch.device('cuda:0')
lib.call('set_conv_stagger', 0)      # 'base' = the plain loop; the 'stagger' columns select the rotated + staggered loop by flag
grid, cin, cout, k = 40, 256, 256, 3
flops = 2.0 * grid ** 3 * cin * cout * k ** 3
for fill in ('randn', 'zeros'):
    x = (torch.randn(1, grid, grid, grid, cin, device=dev) if fill == 'randn' else torch.zeros(1, grid, grid, grid, cin, device=dev)).bfloat16()
    w = torch.randn(cout, cin, k, k, k, device=dev) * (0.05 if fill == 'randn' else 0.0)
    wp, _ = ops.PackedWeight().get([w], torch.bfloat16, cout, False)
    y = torch.empty(1, grid, grid, grid, cout, device=dev, dtype=torch.bfloat16)
    for rnd in range(2):
        line = f'{fill} round {rnd}:'
        for name, fl in (('base', 0), ('stagger', 1024), ('alias', 256), ('nosync', 512), ('alias+nosync', 768), ('stagger+nosync', 1536)):
            t = timeit(lambda: lib.call('conv3d_fwd', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), 1, grid, grid, grid, cin, cout, cout, k, 1, fl, 0, 0,
                                        ops._s()))
            line += f'  {name}: {t*1e3:.1f} us {flops / t / 1e9:.0f} TF'
        print(line, flush=True)
outs = []
for fl in (0, 1024):
    lib.call('conv3d_fwd', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), 1, grid, grid, grid, cin, cout, cout, k, 1, fl, 0, 0, ops._s())
    outs.append(y.clone())
print('stagger == base:', torch.equal(outs[0], outs[1]), flush=True)
x = torch.randn(1, grid, grid, grid, cin, device=dev).bfloat16()
lib.call('set_conv_tile_m', 128)
for kb in (64, 128):
    lib.call('set_conv_kstep_bytes', kb)
    t = timeit(lambda: lib.call('conv3d_fwd', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), 1, grid, grid, grid, cin, cout, cout, k, 1, 0, 0, 0, ops._s()))
    print(f'128x128 tile, K-step {kb} B: {t*1e3:.1f} us {flops / t / 1e9:.0f} TF', flush=True)
lib.call('set_conv_kstep_bytes', 128)
lib.call('set_conv_tile_m', 0)
lib.call('set_conv_stagger', 1)

"""A/B of the K-loop variants of conv_igemm_big_kernel (nrpn_set_conv_stagger 0 = plain loop, 1 = rotated + staggered) on 256->256, 3^3, 40^3 bf16: time on random and zero
operands, and bit-equality of the outputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops  # noqa: E402


def timeit(fn, iters=40, warm=8):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


dev = torch.device('cuda:0')
grid, cin, cout, k = 40, 256, 256, 3
flops = 2.0 * grid ** 3 * cin * cout * k ** 3
outs = {}
for fill in ('randn', 'relu', 'zeros'):
    x = torch.randn(1, grid, grid, grid, cin, device=dev)
    x = {'randn': x, 'relu': x.clamp_min(0), 'zeros': x * 0}[fill].bfloat16()
    w = torch.randn(cout, cin, k, k, k, device=dev) * (0.0 if fill == 'zeros' else 0.05)
    wp, _ = ops.PackedWeight().get([w], torch.bfloat16, cout, False)
    y = torch.empty(1, grid, grid, grid, cout, device=dev, dtype=torch.bfloat16)
    for rnd in range(2):
        line = f'{fill} round {rnd}:'
        for var in (0, 1, 0, 1):
            lib.call('set_conv_stagger', var)
            t = timeit(lambda: lib.call('conv3d_fwd', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), 1, grid, grid, grid, cin, cout, cout, k, 1, 0, 0, 0, ops._s()))
            line += f'  v{var}: {t:.1f} us {flops / t / 1e6:.0f} TF'
            if fill == 'randn':
                outs[var] = y.clone()
        print(line, flush=True)
print('bit-identical outputs:', torch.equal(outs[0], outs[1]))
lib.call('set_conv_stagger', 1)

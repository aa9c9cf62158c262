"""Fixed (prologue + epilogue) vs per-K-step cost of conv_igemm_big_kernel at 256->256 on 40^3, bf16: the 1^3 convolution runs the same
tile grid with 4 K-steps, the 3^3 one with 108:  t = fixed + steps * per_step."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops  # noqa: E402


def timeit(fn, iters=40, warm=5):
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
grid, cin, cout = 40, 256, 256
for fill in ('randn', 'zeros'):
    x = (torch.randn(1, grid, grid, grid, cin, device=dev) if fill == 'randn' else torch.zeros(1, grid, grid, grid, cin, device=dev)).bfloat16()
    y = torch.empty(1, grid, grid, grid, cout, device=dev, dtype=torch.bfloat16)
    t = {}
    for k in (1, 3):
        w = torch.randn(cout, cin, k, k, k, device=dev) * (0.05 if fill == 'randn' else 0.0)
        wp, _ = ops.PackedWeight().get([w], torch.bfloat16, cout, False)
        assert lib.query('conv3d_fwd_plan', 1, grid, grid, grid, cin, cout, k, lib.BF16) == 1, 'not the 256x256 kernel'
        t[k] = timeit(lambda: lib.call('conv3d_fwd', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), 1, grid, grid, grid, cin, cout, cout, k, 1, 0, 0, 0,
                                       ops._s()))
    per = (t[3] - t[1]) / 104
    print(f'{fill}: k1 {t[1]:.1f} us, k3 {t[3]:.1f} us -> per K-step {per:.3f} us, fixed {t[1] - 4 * per:.1f} us '
          f'({100 * (t[1] - 4 * per) / t[3]:.1f} % of the 3^3 launch); MFMA floor per K-step at 2.4 GHz: {2048 / 2.4e3:.3f} us', flush=True)

"""A/B of the 256x256 implicit-GEMM tile on 8 waves (128x64 per wave) vs 4 waves (128x128 per wave, nrpn_conv_opts.tile) on the heavy bf16
shapes, plus the halo-form stem forward vs the im2col form, plus the eval forward with / without the folded BatchNorm (HIP events)."""
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
    return a.elapsed_time(b) / iters * 1e3


dev = torch.device('cuda:0')
torch.manual_seed(0)
SHAPES = [] if os.environ.get('SKIP_TILES') else [(40, 256, 256, 3), (40, 128, 256, 3), (24, 512, 512, 3)]
for grid, cin, cout, k in SHAPES:
    flops = 2.0 * grid ** 3 * cin * cout * k ** 3
    for fill in ('relu', 'randn'):
        x = torch.randn(1, grid, grid, grid, cin, device=dev)
        x = (x.clamp_min(0) if fill == 'relu' else x).bfloat16()
        w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
        wp, _ = ops.PackedWeight().get([w], torch.bfloat16, cout, False)
        line = f'{grid}^3 {cin}->{cout} k{k} {fill}:'
        outs = {}
        for rnd in range(2):
            for tile in ((lib.TILE_256X256, lib.TILE_256X256_W4) + ((lib.TILE_HALO,) if k == 3 else ())):
                t = timeit(lambda: outs.__setitem__(tile, ops._conv_fwd(x, wp, None, cout, cout, k, 0, torch.bfloat16, tile=tile)))
                line += f'  t{tile}: {t:.1f} us {flops / t / 1e6:.0f} TF'
        print(line, ' equal' if torch.equal(outs[lib.TILE_256X256], outs[lib.TILE_256X256_W4]) else ' DIFFERENT',
              ('halo maxdiff %.3g' % (outs[lib.TILE_256X256].float() - outs[lib.TILE_HALO].float()).abs().max().item()) if k == 3 else '', flush=True)

# halo-kernel variants (nrpn_conv_opts.debug bits 12-13)
x = torch.randn(1, 40, 40, 40, 256, device=dev).clamp_min(0).bfloat16()
w = torch.randn(256, 256, 3, 3, 3, device=dev) * 0.05
wp, _ = ops.PackedWeight().get([w], torch.bfloat16, 256, False)
y = torch.empty(1, 40, 40, 40, 256, device=dev, dtype=torch.bfloat16)
for rnd in range(2):
    line = 'halo variants 256->256@40^3:'
    for var in (0, 1, 2, 3):
        o = lib.ConvOpts(tile=lib.TILE_HALO, debug=var << 12)
        t = timeit(lambda: lib.call('conv3d_fwd_ex', x.data_ptr(), wp.data_ptr(), 0, y.data_ptr(), 1, 40, 40, 40, 256, 256, 256, 3, lib.BF16, 0, 0, o.ptr(), ops._s()), iters=40)
        line += f'  V{var}: {t:.1f} us {2 * 64000 * 256 * 256 * 27 / t / 1e6:.0f} TF'
    print(line, flush=True)
if os.environ.get('SKIP_REST'):
    sys.exit(0)
# stem
from torch import nn  # noqa: E402
from nerf_rpn_amd.model import hip_nn  # noqa: E402
conv = nn.Conv3d(4, 64, 7, stride=2, padding=3).to(dev)
x = torch.rand(1, 160, 160, 160, 4, device=dev).bfloat16()
with torch.no_grad():
    for halo in (True, False, True, False):
        ops.STEM_HALO[0] = halo
        t = timeit(lambda: hip_nn.conv3d(conv, x))
        print(f'stem 160^3 halo={halo}: {t:.1f} us {89.9e9 / t / 1e6:.0f} TF(useful)', flush=True)
ops.STEM_HALO[0] = True

# eval forward of VGG19 + FPN with / without the folded BatchNorm
import bench  # noqa: E402
model = bench.build_model(torch.bfloat16, dev).eval()
xs = torch.rand(4, 160, 160, 160, device=dev)
with torch.no_grad():
    for fold, halo in ((True, True), (False, True), (True, False), (False, False), (True, True)):
        hip_nn.FOLD_EVAL_BN, ops.STEM_HALO[0] = fold, halo
        t = timeit(lambda: model.backbone(xs.unsqueeze(0)), iters=10, warm=3)
        print(f'eval forward VGG19+FPN fold={fold} halo={halo}: {t / 1e3:.3f} ms  {1713.2 / (t / 1e3):.0f} TF = {1713.2 / (t / 1e3) / 2500:.3f} of peak', flush=True)
    hip_nn.FOLD_EVAL_BN, ops.STEM_HALO[0] = True, True
    for tile in (lib.TILE_256X256, lib.TILE_HALO, 0):
        ops.CONV_TILE[0] = tile
        t = timeit(lambda: model.backbone(xs.unsqueeze(0)), iters=10, warm=3)
        print(f'eval forward VGG19+FPN tile={tile}: {t / 1e3:.3f} ms = {1713.2 / (t / 1e3) / 2500:.3f} of peak', flush=True)
ops.CONV_TILE[0] = 0

"""Round-5 kernel A/Bs on one box (HIP events, alternating rounds):
  (a) halo kernel, classic K order (14 K-steps per chunk) vs cross-chunk tap pairing (nrpn_conv_opts.halo_pairing), 256 -> 256 @ 40^3 and
      128 -> 256 @ 40^3, post-ReLU and random operands; outputs must be bit-identical;
  (b) the bf16x3 building blocks: split pass, halo kernel with tripled K and fp32 rows, bf16x3 wgrad over three planes vs the fp32 kernels;
  (c) the HBM-bound companions on their largest shapes are in bench.py's roofline.hbm_stages."""
import json
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
out = {"halo_pairing": [], "bf16x3": []}
for grid, cin, cout in ((40, 256, 256), (40, 128, 256)):
    flops = 2.0 * grid ** 3 * cin * cout * 27
    for fill in ('relu', 'randn'):
        x = torch.randn(1, grid, grid, grid, cin, device=dev)
        x = (x.clamp_min(0) if fill == 'relu' else x).bfloat16()
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
        wp, _ = ops.PackedWeight().get([w], torch.bfloat16, cout, False)
        res, ts = {}, {1: [], 2: []}
        for rnd in range(3):
            for pairing in (2, 1):
                ts[pairing].append(timeit(lambda: res.__setitem__(pairing, ops._conv_fwd(x, wp, None, cout, cout, 3, 0, torch.bfloat16, tile=lib.TILE_HALO,
                                                                                          halo_pairing=pairing)), iters=40))
        row = {"shape": f"{cin}->{cout}@{grid}^3", "fill": fill, "classic_us": [round(t, 1) for t in ts[2]], "paired_us": [round(t, 1) for t in ts[1]],
               "classic_tflops": round(flops / min(ts[2]) / 1e6, 1), "paired_tflops": round(flops / min(ts[1]) / 1e6, 1),
               "bit_identical": bool(torch.equal(res[1], res[2]))}
        out["halo_pairing"].append(row)
        print(row, flush=True)

# bf16x3 building blocks on the dominant layer
grid, c = 40, 256
xf = torch.randn(1, grid, grid, grid, c, device=dev).clamp_min(0)
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
wp32, _ = ops.PackedWeight().get([w], torch.float32, c, False)
w3 = ops.split3(wp32, 0b010)[0]
flops = 2.0 * grid ** 3 * c * c * 27
t_split = timeit(lambda: ops.split3(xf, 0b100))
xs = ops.split3(xf, 0b100)[0]
for pairing in (2, 1):
    t3 = timeit(lambda: ops._conv_fwd(xs, w3, None, c, c, 3, 0, torch.float32, halo_pairing=pairing), iters=20)
    out["bf16x3"].append({"what": f"fwd 256->256@40^3 bf16x3 (K tripled, fp32 rows), halo_pairing={pairing}", "us": round(t3, 1),
                          "fp32_equivalent_tflops": round(flops / t3 / 1e6, 1), "executed_tflops": round(3 * flops / t3 / 1e6, 1)})
t32 = timeit(lambda: ops._conv_fwd(xf, wp32, None, c, c, 3, 0, torch.float32), iters=5, warm=2)
y3 = ops._conv_fwd(xs, w3, None, c, c, 3, 0, torch.float32)
y32 = ops._conv_fwd(xf, wp32, None, c, c, 3, 0, torch.float32)
out["bf16x3"].append({"what": "fwd 256->256@40^3 fp32 MFMA kernel", "us": round(t32, 1), "tflops": round(flops / t32 / 1e6, 1)})
out["bf16x3"].append({"what": "split pass 256@64000 (f32 -> interleaved bf16 [3C])", "us": round(t_split, 1),
                      "gbs": round((xf.numel() * 10) / t_split / 1e3, 1)})
out["bf16x3"].append({"what": "bf16x3 vs fp32 kernel, max |diff| / max |y|", "value": float((y3 - y32).abs().max() / y32.abs().max())})
# wgrad: three planes on the batch axis vs the fp32 wgrad kernel
dy = torch.randn(1, grid, grid, grid, c, device=dev)
xp, dyp = ops.split3(xf, None, 3, 0b010)[1], ops.split3(dy, None, 3, 0b100)[1]


def wgrad(xx, dd, n, code):
    slices = ops.query("conv3d_wgrad_slices", n, grid, grid, grid, c, c, c, 3, code)
    gwp = torch.empty((slices, 27, c, c), dtype=torch.float32, device=dev)
    ws = torch.empty(ops.query("conv3d_wgrad_workspace_bytes", n, grid, grid, grid, c, c, c, 3, code), dtype=torch.uint8, device=dev)
    lib.call("conv3d_wgrad", xx.data_ptr(), dd.data_ptr(), gwp.data_ptr(), 0, n, grid, grid, grid, c, c, c, 3, code, 0, ws.data_ptr(), ops._s())
    return gwp.sum(0)


tw3 = timeit(lambda: wgrad(xp, dyp, 3, lib.BF16), iters=10, warm=2)
tw32 = timeit(lambda: wgrad(xf, dy, 1, lib.F32), iters=4, warm=1)
g3, g32 = wgrad(xp, dyp, 3, lib.BF16), wgrad(xf, dy, 1, lib.F32)
out["bf16x3"].append({"what": "wgrad 256x256@40^3: bf16x3 (3 planes) vs fp32 kernel", "bf16x3_us": round(tw3, 1), "fp32_us": round(tw32, 1),
                      "max_rel_diff": float((g3 - g32).abs().max() / g32.abs().max())})
for r in out["bf16x3"]:
    print(r, flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None

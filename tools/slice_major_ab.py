"""A/B of the XCD order of K-sliced 128-row conv launches (tools switch nrpn_set_conv_slice_major: 0 = M-tile-major, 2 = slice-major) on the
pyramid-level shapes of the headline model; checks that both orders give the same bits."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from nerf_rpn_amd import lib, ops

if not hasattr(lib.load(), 'nrpn_set_conv_slice_major'):
    sys.exit('this A/B needs the experimental switch nrpn_set_conv_slice_major: git apply tools/patches/r5_slice_major_xcd_order.patch && make -C nerf_rpn_amd/csrc')
dev = torch.device('cuda:0')
dtype = torch.bfloat16
out = []
for (g, cin, cout) in ((10, 512, 512), (5, 512, 512), (10, 256, 256), (5, 256, 256), (10, 512, 256), (20, 256, 256), (10, 256, 512)):
    torch.manual_seed(0)
    x = torch.randn(1, g, g, g, cin, device=dev).clamp_min(0).to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    wp, _ = ops.PackedWeight().get([w], dtype, cout, True)
    row = {"shape": f"{cin}->{cout}@{g}^3", "ksplit": lib.query('conv3d_fwd_workspace_bytes', 1, g, g, g, cin, cout, 3, 1) // (4 * g ** 3 * cout)}
    ys = {}
    for mode in (0, 2, 1):
        lib.call('set_conv_slice_major', mode)
        for _ in range(5):
            y = ops._conv_fwd(x, wp, None, cout, cout, 3, 0, dtype)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            y = ops._conv_fwd(x, wp, None, cout, cout, 3, 0, dtype)
        b.record()
        torch.cuda.synchronize()
        row[{0: "m_major_us", 2: "slice_major_us", 1: "default_us"}[mode]] = round(1e3 * a.elapsed_time(b) / 100, 2)
        ys[mode] = y.clone()
    row["same_bits"] = bool(torch.equal(ys[0], ys[2]) and torch.equal(ys[0], ys[1]))
    out.append(row)
    print(row, flush=True)
lib.call('set_conv_slice_major', 1)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)

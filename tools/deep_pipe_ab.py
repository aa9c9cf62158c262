"""A/B of the LDS ring depth of K-sliced 128-row conv launches (tools switch nrpn_set_conv_deep_pipe: 0 = two stages / ~384 workgroups, N = four
stages / <= 256 workgroups for launches of <= N tiles) on the pyramid-level shapes of the headline model, against an fp32 torch reference."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from nerf_rpn_amd import lib, ops

if not hasattr(lib.load(), 'nrpn_set_conv_deep_pipe'):
    sys.exit('this A/B needs the experimental switch nrpn_set_conv_deep_pipe: git apply tools/patches/r5_deep_pipe_ksliced.patch && make -C nerf_rpn_amd/csrc')
dev = torch.device('cuda:0')
dtype = torch.bfloat16
out = []
for (g, cin, cout) in ((10, 512, 512), (5, 512, 512), (10, 256, 256), (5, 256, 256), (10, 512, 256), (20, 256, 256), (10, 256, 512), (20, 128, 128)):
    torch.manual_seed(0)
    x = torch.randn(1, g, g, g, cin, device=dev).clamp_min(0).to(dtype)
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).to(dtype).float()
    wp, _ = ops.PackedWeight().get([w], dtype, cout, True)
    ref = F.conv3d(x.float().permute(0, 4, 1, 2, 3), w, padding=1).permute(0, 2, 3, 4, 1)
    row = {"shape": f"{cin}->{cout}@{g}^3"}
    for mode in (0, 64, 128):
        lib.call('set_conv_deep_pipe', mode)
        row[f"slices_{mode}"] = lib.query('conv3d_fwd_workspace_bytes', 1, g, g, g, cin, cout, 3, 1) // (4 * g ** 3 * cout)
        for _ in range(5):
            y = ops._conv_fwd(x, wp, None, cout, cout, 3, 0, dtype)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            y = ops._conv_fwd(x, wp, None, cout, cout, 3, 0, dtype)
        b.record()
        torch.cuda.synchronize()
        row[f"us_{mode}"] = round(1e3 * a.elapsed_time(b) / 100, 2)
        row[f"err_{mode}"] = float((y.float() - ref).abs().max() / ref.abs().max())
    out.append(row)
    print(row, flush=True)
lib.call('set_conv_deep_pipe', 128)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)

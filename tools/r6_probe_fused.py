"""NEEDS tools/patches/r6_fused_splitk.patch applied (git apply) and the library rebuilt: ops.FUSED_SPLIT / NRPN_FUSED_DBG exist only there.
EXPERIMENT: time the K-sliced 20^3 launches in the fused / two-launch forms (and the fused form's debug variants)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops
dev = torch.device('cuda:0')
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (grid, cin, cout) in [((20, 20, 20), 512, 512), ((20, 20, 20), 256, 512)]:
    x = torch.randn(1, *grid, cin, device=dev).relu().bfloat16()
    w = (torch.randn(27, cout, cin, device=dev) * 0.05).bfloat16()
    res = {}
    for fused in (False, True):
        ops.FUSED_SPLIT[0] = fused
        res[fused] = timeit(lambda: ops._conv_fwd(x, w, None, cout, cout, 3, 0, torch.bfloat16))
    ops.FUSED_SPLIT[0] = True
    print(grid, cin, cout, f"dbg={os.environ.get('NRPN_FUSED_DBG', '0')}: two-launch {res[False]:.1f} us, fused {res[True]:.1f} us", flush=True)

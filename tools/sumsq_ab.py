"""A/B of nrpn_grad_sumsq's read pattern (tools switch nrpn_set_sumsq_form) on the 74.8 M-float gradient arena of the headline model."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_rpn_amd import lib, ops

dev = torch.device("cuda:0")
n = 74_815_925
g = torch.randn(n, device=dev)
ss = torch.zeros(lib.query("grad_sumsq_floats"), device=dev)
ref = float((g.double() ** 2).sum())
out = []
for form in (0, 1, 2):
    for grid in (512, 1024, 2048):
        lib.call("set_sumsq_form", form, grid)
        for _ in range(5):
            ops.grad_sumsq(g, ss, 1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            ops.grad_sumsq(g, ss, 1.0)
        b.record()
        torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 50
        out.append({"form": form, "grid": grid, "avg_us": round(us, 1), "gbs": round(4 * n / us / 1e3, 1), "rel_err": abs(float(ss[0]) - ref) / ref})
        print(out[-1], flush=True)
lib.call("set_sumsq_form", 0, 1024)
json.dump(out, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None

"""Per-parameter gradient error of the HIP path against a train_* fixture: err / allowed (test_gpu_e2e.test_train_matches_reference's rule),
the reference's own fp32 error (err32) and the fp64 scale.   python tools/diag_train_fixture.py <fixture name>"""
import os
import sys

import numpy as np
import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from test_gpu_e2e import T, build, scene  # noqa: E402

name = sys.argv[1]
dev = torch.device("cuda", 0)
g = np.load(os.path.join(root, "tests", "golden", name + ".npz"), allow_pickle=True)
rot = bool(g["rotated"])
m = build(rot, 160, dev, str(g["reg_loss_type"]), backbone=str(g["backbone"]) if "backbone" in g else "vgg", sd=0.0).train()
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    m.set_compute_dtype(torch.bfloat16)
xs = [scene(s, 200 + i).to(dev) for i, s in enumerate(g["shapes"])]
gts = [T(g[f"gt{i}"], dev) for i in range(len(xs))]
pos, neg = T(g["pos_idx"], dev), T(g["neg_idx"], dev)
m.rpn.sampler_hook = lambda labels: (pos, neg)
_, losses, _ = m(xs, gts)
print({k: (v.item(), float(g[k])) for k, v in losses.items()})
(losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]).backward()
params = dict(m.backbone.named_parameters())
params.update({"head." + k: v for k, v in m.rpn.head.named_parameters()})
rows = []
for k, p in params.items():
    if "grad/" + k in g:
        ref, got = T(g["grad/" + k]), p.grad.cpu()
        r64 = T(g["grad64/" + k])
    else:
        ref, got = T(g["gval/" + k]), p.grad.reshape(-1)[T(g["gidx/" + k], dev)].cpu()
        r64 = T(g["gval64/" + k])
    scale = float(g["gmax64/" + k])
    err = (got - ref).abs().max().item()
    err64 = (got.double() - r64.double()).abs().max().item()
    allowed = max(0.1 * scale, 4.0 * float(g["err32/" + k])) + 5e-5
    ev = (got - ref).abs().reshape(-1)
    top = torch.topk(ev, min(4, ev.numel())).values.tolist()
    rows.append((err / allowed, k, err, err64, float(g["err32/" + k]), scale, tuple(p.shape), int((ev > allowed).sum()), ev.numel(), [round(t, 4) for t in top]))
fr, fg = [], []
for k, p in params.items():
    scale = float(g["gmax64/" + k])
    if scale > 1e-6:
        if "grad/" + k in g:
            ref, got = T(g["grad/" + k]), p.grad.float().cpu()
        else:
            ref, got = T(g["gval/" + k]), p.grad.float().reshape(-1)[T(g["gidx/" + k], dev)].cpu()
        fr.append(ref.reshape(-1) / scale); fg.append(got.reshape(-1) / scale)
a, b = torch.cat(fr).double(), torch.cat(fg).double()
print("whole-gradient cosine", (a @ b / (a.norm() * b.norm())).item(), "norm ratio", (b.norm() / a.norm()).item())
per = []
for k, p in params.items():
    if "gnorm/" + k in g and float(g["gnorm/" + k]) > 0:
        per.append((p.grad.float().norm().item() / float(g["gnorm/" + k]), k))
per.sort()
print("per-parameter gradient-norm ratio: min", per[0], "median", per[len(per) // 2], "max", per[-1])
cosines = []
for k, p in params.items():
    if "gidx/" + k in g and float(g["gmax64/" + k]) > 1e-6 and k.endswith("weight") and p.dim() > 1:
        ref, got = T(g["gval/" + k]).double(), p.grad.float().reshape(-1)[T(g["gidx/" + k], dev)].cpu().double()
        cosines.append(((ref @ got / (ref.norm() * got.norm() + 1e-30)).item(), k, p.grad.float().norm().item() / float(g["gnorm/" + k])))
cosines.sort()
print("GEMM weights: per-tensor cosine on the 256 sampled entries: min", cosines[0][:2], "p10", cosines[len(cosines) // 10][:2], "median", cosines[len(cosines) // 2][:2])
rs = sorted(c[2] for c in cosines)
print("GEMM weights: gradient-norm ratio min / median / max", rs[0], rs[len(rs) // 2], rs[-1], "count", len(rs))
rows.sort(reverse=True)
for r in rows[:14]:
    print(f"{r[0]:6.2f}  {r[1]:40s} err vs ref {r[2]:.4g}  vs fp64 {r[3]:.4g}  ref's own err32 {r[4]:.4g}  scale {r[5]:.4g}  {r[6]}  over the bound: {r[7]}/{r[8]}  largest errors {r[9]}")

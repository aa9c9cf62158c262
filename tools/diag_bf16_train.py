"""Where does the bf16 training gradient leave the fp32 one?  Same HIP model, same input, same sampled anchors, fp32 vs bf16 compute:
relative error of the FPN outputs (train-mode BatchNorm), the losses, and the cosine of every GEMM weight's gradient in network order.
    python tools/diag_bf16_train.py <train fixture>"""
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
out = {}
for dt in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    m = build(rot, 160, dev, str(g["reg_loss_type"]), backbone=str(g["backbone"]) if "backbone" in g else "vgg", sd=0.0).train()
    m.set_compute_dtype(dt)
    xs = [scene(s, 200 + i).to(dev) for i, s in enumerate(g["shapes"])]
    gts = [T(g[f"gt{i}"], dev) for i in range(len(xs))]
    pos, neg = T(g["pos_idx"], dev), T(g["neg_idx"], dev)
    m.rpn.sampler_hook = lambda labels: (pos, neg)
    (feats, _, _), losses, _ = m(xs, gts)
    (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]).backward()
    params = dict(m.backbone.named_parameters())
    params.update({"head." + k: v for k, v in m.rpn.head.named_parameters()})
    out[dt] = ([f.detach().float() for f in feats], {k: v.item() for k, v in losses.items()}, {k: p.grad.detach().float().clone() for k, p in params.items()})
f32, b16 = out[torch.float32], out[torch.bfloat16]
for i, (a, b) in enumerate(zip(f32[0], b16[0])):
    print(f"FPN output {i} {tuple(a.shape)}: max |diff| / max |fp32| = {(a - b).abs().max().item() / a.abs().max().item():.4f}, rms diff / rms = {((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()).item():.4f}")
print("losses fp32", f32[1], "bf16", b16[1])
for k in f32[2]:
    a, b = f32[2][k].reshape(-1).double(), b16[2][k].reshape(-1).double()
    if f32[2][k].dim() > 1:
        print(f"  {k:44s} cos {(a @ b / (a.norm() * b.norm() + 1e-30)).item():7.4f}  norm ratio {(b.norm() / (a.norm() + 1e-30)).item():7.4f}")

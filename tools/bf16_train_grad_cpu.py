"""bf16-storage emulation of a TRAINING pass of the oracle detector on the CPU (forward outputs and backward grad-inputs of every conv /
norm / pool / activation rounded to bf16, bf16-rounded GEMM weights, fp32 accumulation and fp32 weight gradients), against the same pass in
fp32: per GEMM weight, cosine and norm ratio of the gradient.  Says how much of the fp32 gradient direction survives bf16 storage in these
randomly initialised, batch-1, train-mode-BatchNorm networks -- the figure the HIP bf16 path is held to.
    python tools/bf16_train_grad_cpu.py <train fixture> [...]      appends to tests/golden/bf16_emulation.json under 'grad/<fixture>'"""
import json
import os
import sys

import numpy as np
import torch
from torch import nn

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from fixture_init import seeded_state  # noqa: E402
from oracle import nets as ON, rpn as OR  # noqa: E402

ROUNDED = (nn.Conv3d, nn.BatchNorm3d, nn.MaxPool3d, nn.ReLU)


def build(g):
    bbk, rot = str(g["backbone"]) if "backbone" in g else "vgg", bool(g["rotated"])
    bb = ON.ResNetFPN() if bbk == "resnet" else ON.VGGFPN("EF", 4, 160)
    hd = ON.RPNHead(256, 13, 4, rot)
    seeded_state(bb, 1)
    seeded_state(hd, 2)
    det = OR.Detector(bb, OR.RPN(hd, rotated=rot, reg_loss_type=str(g["reg_loss_type"]), pre_nms_top_n=2500, post_nms_top_n=2500))
    bb.train()
    hd.train()
    return bb, hd, det


def one(g, bf16):
    bb, hd, det = build(g)
    if bf16:
        with torch.no_grad():
            for p in list(bb.parameters()) + list(hd.parameters()):
                if p.dim() > 1:
                    p.copy_(p.bfloat16().float())
        fw = lambda mod, inp, out: out.bfloat16().float()
        bw = lambda mod, gin, gout: tuple(None if t is None else t.bfloat16().float() for t in gin)
        mods = [m for m in bb.modules() if isinstance(m, ROUNDED)] + [m for m in hd.conv.modules() if isinstance(m, (nn.Conv3d, nn.ReLU))]
        first = next(m for m in bb.modules() if isinstance(m, nn.Conv3d))      # the stem: its input needs no gradient, nothing to round
        for m in mods:
            if isinstance(m, nn.ReLU):
                m.inplace = False
            m.register_forward_hook(fw)
            if m is not first:
                m.register_full_backward_hook(bw)
    xs = [torch.rand(4, *[int(v) for v in s], generator=torch.Generator().manual_seed(200 + i)) for i, s in enumerate(g["shapes"])]
    gts = [torch.from_numpy(g[f"gt{i}"]) for i in range(len(xs))]
    pos, neg = torch.from_numpy(g["pos_idx"]), torch.from_numpy(g["neg_idx"])
    det.rpn.sampler_hook = lambda labels: (pos, neg)
    _, losses, _, _ = det([x.bfloat16().float() if bf16 else x for x in xs], gts, training=True)
    (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]).backward()
    params = dict(bb.named_parameters())
    params.update({"head." + k: v for k, v in hd.named_parameters()})
    return {k: v.item() for k, v in losses.items()}, {k: p.grad.detach().reshape(-1).double() for k, p in params.items() if p.dim() > 1}


path = os.path.join(root, "tests", "golden", "bf16_emulation.json")
res = json.load(open(path))
for name in sys.argv[1:]:
    g = np.load(os.path.join(root, "tests", "golden", name + ".npz"), allow_pickle=True)
    l32, g32 = one(g, False)
    l16, g16 = one(g, True)
    live = [k for k, a in g32.items() if a.norm() > 0 and g16[k].norm() > 0]      # (a level no sampled anchor falls on has no gradient)
    cos = sorted((g32[k] @ g16[k] / (g32[k].norm() * g16[k].norm())).item() for k in live)
    ratio = sorted((g16[k].norm() / g32[k].norm()).item() for k in live)
    res["grad/" + name] = {"cos_min": round(cos[0], 4), "cos_p10": round(cos[len(cos) // 10], 4), "cos_median": round(cos[len(cos) // 2], 4),
                           "cos_max": round(cos[-1], 4), "norm_ratio_min": round(ratio[0], 4), "norm_ratio_max": round(ratio[-1], 4),
                           "loss_fp32": l32, "loss_bf16": l16, "tensors": len(cos)}
    print(name, res["grad/" + name], flush=True)
json.dump(res, open(path, "w"), indent=1)

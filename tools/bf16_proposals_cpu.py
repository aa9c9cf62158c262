"""What bf16 STORAGE alone does to the proposals of the full-size eval fixtures: the oracle detector (plain torch on the CPU, the fixtures'
weights) run with input, weights and every backbone / head-conv output rounded to bf16 (fp32 accumulation; the head's output GEMM stays
fp32, as on the GPU), compared with the reference's fp32 proposals stored in the fixture exactly as tests/test_gpu_fullsize.py compares the
HIP bf16 run: fraction of the reference's top-300 proposals that have an emulated proposal with IoU above 0.9 / 0.7 / 0.5.
    python tools/bf16_proposals_cpu.py         writes tests/golden/bf16_emulation_proposals.json"""
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
from oracle import boxes as OB, nets as ON, rpn as OR  # noqa: E402

SWIN_S = dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24))
ROUNDED = (nn.Conv3d, nn.BatchNorm3d, nn.MaxPool3d, nn.ReLU, nn.Linear, nn.LayerNorm, nn.GELU, ON.WindowAttention, ON.SwinBlock)


def build(g):
    bbk, rot = str(g["backbone"]) if "backbone" in g else "vgg", bool(g["rotated"])
    if bbk == "swin":
        bb = ON.SwinFPN(SWIN_S["embed_dim"], SWIN_S["depths"], SWIN_S["num_heads"], 0.1)
    else:
        bb = ON.ResNetFPN() if bbk == "resnet" else ON.VGGFPN("EF", 4, 160)
    hd = ON.RPNHead(256, 13, 4, rot)
    seeded_state(bb, 1)
    seeded_state(hd, 2)
    bb.eval()
    hd.eval()
    return bb, hd, OR.Detector(bb, OR.RPN(hd, rotated=rot, pre_nms_top_n=2500, post_nms_top_n=2500)), rot


res = {"note": "fraction of the reference's fp32 top-300 proposals (fixture) matched by a proposal of the bf16-storage emulation of the oracle "
               "detector on the CPU (tools/bf16_proposals_cpu.py) at IoU > 0.9 / 0.7 / 0.5, and the emulated score range"}
names = sys.argv[1:] or ["eval_resnet_aabb_160x120x64", "eval_swin_obb_160x120x64", "eval_resnet_obb_200x200x130", "eval_swin_obb_200x200x130"]
for name in names:
    g = np.load(os.path.join(root, "tests", "golden", name + ".npz"), allow_pickle=True)
    bb, hd, det, rot = build(g)
    shape = [int(s) for s in (g["shape"] if "shape" in g else g["shapes"][0])]
    seed = int(g["seed"]) if "seed" in g else 100
    if "normalize_density" in g:       # the VGG19 full-size fixtures: raw [W, L, H, 4] scene through the loader's ingest (density -> alpha)
        gen = torch.Generator().manual_seed(seed)
        raw = torch.rand(*shape, 4, generator=gen)
        raw[..., 3] = raw[..., 3] * 10.0 - 5.0
        if bool(g["normalize_density"]):
            raw[..., 3] = torch.clamp(1.0 - torch.exp(-torch.exp(raw[..., 3]) / 100.0), 0.0, 1.0)
        x = raw.permute(3, 0, 1, 2).contiguous()
    else:
        x = torch.rand(4, *shape, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        for p in list(bb.parameters()) + list(hd.parameters()):
            if p.dim() > 1:
                p.copy_(p.bfloat16().float())
    rnd = lambda mod, inp, out: out.bfloat16().float()
    for mod in bb.modules():
        if isinstance(mod, ROUNDED):
            mod.register_forward_hook(rnd)
    for mod in hd.conv.modules():
        if isinstance(mod, (nn.Conv3d, nn.ReLU)):
            mod.register_forward_hook(rnd)
    with torch.no_grad():
        (feats, props, lvls), _, scores, aux = det([x.bfloat16().float()])
    rp, gp = torch.from_numpy(g["proposals0"])[:300], props[0].float()
    iou = OB.iou_matrix(rp, gp) if rot else OB.aabb_iou_matrix(rp, gp)
    best = iou.max(dim=1).values
    res[name] = {"matched_0.9": round((best > 0.9).float().mean().item(), 4), "matched_0.7": round((best > 0.7).float().mean().item(), 4),
                 "matched_0.5": round((best > 0.5).float().mean().item(), 4), "proposals": int(gp.shape[0]),
                 "scores": [round(float(scores[0].min()), 4), round(float(scores[0].max()), 4)]}
    print(name, res[name], flush=True)
path = os.path.join(root, "tests", "golden", "bf16_emulation_proposals.json")
if os.path.exists(path):
    old = json.load(open(path))
    old.update(res)
    res = old
json.dump(res, open(path, "w"), indent=1)

"""bf16-storage emulation of the oracle Swin-S + FCOS detector on the CPU (see tools/bf16_proposals_cpu.py), full-size FCOS fixtures:
fraction of the reference's top-300 detections matched by an emulated detection at rotated IoU > 0.9 / 0.7 / 0.5.
    python tools/bf16_fcos_cpu.py        appends to tests/golden/bf16_emulation_proposals.json"""
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
from oracle import boxes as OB, fcos as OF, nets as ON  # noqa: E402

SWIN_S = dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24))
ROUNDED = (nn.Conv3d, nn.BatchNorm3d, nn.MaxPool3d, nn.ReLU, nn.Linear, nn.LayerNorm, nn.GELU, nn.GroupNorm, ON.WindowAttention, ON.SwinBlock)
OUT_GEMMS = ("cls_logits", "bbox_pred", "centerness")        # fp32 outputs on the GPU as well (FcosHeadOutFn reads the f32 GEMM rows)


def build(rot):
    ob = ON.SwinFPN(SWIN_S["embed_dim"], SWIN_S["depths"], SWIN_S["num_heads"], 0.0)
    seeded_state(ob, 1)
    oh = OF.FCOSHead(256, 4, [4, 8, 16, 32], True, True, rot)
    seeded_state(oh, 2, bias_jitter=0.5)
    for l, sc in enumerate(oh.scales):
        sc.scale.data.fill_(0.8 + 0.15 * l)
    ob.eval()
    oh.eval()
    return ob, oh, OF.FCOS(ob, oh, [4, 8, 16, 32], rot, 1.5, "iou", True, False, 0.0, 0.0, 2500, 0.3, 2500, 0.0)


path = os.path.join(root, "tests", "golden", "bf16_emulation_proposals.json")
res = json.load(open(path)) if os.path.exists(path) else {}
for name in sys.argv[1:] or ["fcos_eval_obb_swin_160x120x64", "fcos_eval_obb_swin_200x200x130"]:
    g = np.load(os.path.join(root, "tests", "golden", name + ".npz"), allow_pickle=True)
    rot = bool(g["rotated"])
    x = torch.rand(4, *[int(s) for s in g["shape"]], generator=torch.Generator().manual_seed(int(g["seed"])))
    out = {}
    for mode in ("fp32", "bf16"):
        ob, oh, det = build(rot)
        if mode == "bf16":
            with torch.no_grad():
                for p in list(ob.parameters()) + list(oh.parameters()):
                    if p.dim() > 1:
                        p.copy_(p.bfloat16().float())
            rnd = lambda mod, inp, out_: out_.bfloat16().float()
            for mod in ob.modules():
                if isinstance(mod, ROUNDED):
                    mod.register_forward_hook(rnd)
            for nm, mod in oh.named_modules():
                if isinstance(mod, ROUNDED) and not any(nm.startswith(k) or ("." + k) in nm for k in OUT_GEMMS):
                    mod.register_forward_hook(rnd)
        with torch.no_grad():
            boxes, _, scores, aux = det([x.bfloat16().float() if mode == "bf16" else x])
        out[mode] = (boxes[0].float(), scores[0].float())
    rp, rs = torch.from_numpy(g["boxes0"]), torch.from_numpy(g["scores0"])
    assert out["fp32"][0].shape == rp.shape and (out["fp32"][0] - rp).abs().max() < 5e-2, "the oracle does not reproduce the fixture in fp32"
    top = torch.argsort(rs, descending=True, stable=True)[:300]
    ref_boxes = rp[top][:, -7:] if rot else rp[top][:, -6:]
    got_boxes = out["bf16"][0][:, -7:] if rot else out["bf16"][0][:, -6:]
    iou = OB.iou_matrix(ref_boxes, got_boxes) if rot else OB.aabb_iou_matrix(ref_boxes, got_boxes)
    best = iou.max(dim=1).values
    res[name] = {"matched_0.9": round((best > 0.9).float().mean().item(), 4), "matched_0.7": round((best > 0.7).float().mean().item(), 4),
                 "matched_0.5": round((best > 0.5).float().mean().item(), 4), "proposals": int(got_boxes.shape[0]),
                 "scores": [round(float(out["bf16"][1].min()), 4), round(float(out["bf16"][1].max()), 4)]}
    print(name, res[name], "fixture boxes", tuple(rp.shape), flush=True)
json.dump(res, open(path, "w"), indent=1)

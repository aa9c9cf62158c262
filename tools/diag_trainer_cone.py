"""Which parameters differ between FlatTrainer and torch.optim.AdamW after 3 steps (tests/test_gpu_trainer.py), cone on / off."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_e2e import T, build, scene
from nerf_rpn_amd import ops
from nerf_rpn_amd.engine import FlatTrainer
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "train_obb.npz")))
xs = [scene(s, 200 + i).to(dev) for i, s in enumerate(g["shapes"])]
gts = [T(g[f"gt{i}"], dev) for i in range(len(xs))]
pos, neg = T(g["pos_idx"], dev), T(g["neg_idx"], dev)
def loss_of(m):
    m.rpn.sampler_hook = lambda labels: (pos, neg)
    _, l, _ = m(xs, gts)
    return l["loss_objectness"] + 5.0 * l["loss_rpn_box_reg"]
for cone in (True,):
    ops.CONE_ENABLED[0] = cone
    lr, steps = 3e-4, 3
    ref = build(True, 160, dev).train()
    opt = torch.optim.AdamW(ref.parameters(), lr=lr, weight_decay=0.01)
    gref, wref = [], []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss_of(ref).backward()
        gref.append({k: p.grad.clone() for k, p in ref.named_parameters()})
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        opt.step()
        wref.append({k: p.detach().clone() for k, p in ref.named_parameters()})
    m = build(True, 160, dev).train()
    tr = FlatTrainer(m, lr=lr, weight_decay=0.01, clip_grad_norm=0.1)
    for s in range(steps):
        loss_of(m).backward()
        torch.cuda.synchronize()
        worst = max(((p.grad - gref[s][k]).abs().max().item() / (gref[s][k].abs().max().item() + 1e-30), k) for k, p in m.named_parameters())
        print(f"cone={cone} step {s}: worst relative gradient difference trainer vs autograd {worst[0]:.3e} at {worst[1]}")
        for k in ("backbone.fpn_neck.fpn_convs.0.weight", "backbone.fpn_neck.fpn_convs.1.weight", "backbone.fpn_neck.fpn_convs.2.weight", "backbone.fpn_neck.fpn_convs.3.weight", "rpn.head.conv.0.weight", "rpn.head.conv.2.weight", "rpn.head.conv.4.weight", "rpn.head.conv.6.weight", "rpn.head.conv.0.bias", "rpn.head.cls_logits.weight"):
            pg = dict(m.named_parameters())[k].grad
            print(f"      {k}: rel diff {(pg - gref[s][k]).abs().max().item() / gref[s][k].abs().max().item():.3e}  |g|max {gref[s][k].abs().max().item():.3e}")
        tr.step()
        torch.cuda.synchronize()
        wd = max(((p.detach() - wref[s][k]).abs().max().item(), k) for k, p in m.named_parameters() if not k.endswith("bias"))
        print(f"      weights after step {s}: worst abs difference (non-bias) {wd[0]:.3e} at {wd[1]}")
    rows = []
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        gs = torch.stack([gr[k].abs() > 1e-2 * gr[k].abs().max() for gr in gref]).all(dim=0)
        d = (p.detach() - q.detach()).abs()
        rows.append((d[gs].max().item() if gs.any() else 0.0, k))
    rows.sort(reverse=True)
    print(f"cone={cone}: weight differences on significant entries (lr={lr}):", [(f"{a:.2e}", k) for a, k in rows[:6]])

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_fcos import build, scene, T
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "fcos_eval_obb_batch2"
g = dict(np.load(os.path.join(ROOT, "tests/golden", name + ".npz")))
rot = bool(g["rotated"])
m = build(rot, str(g["backbone"]), dev, pre_nms_top_n=int(g["pre_nms_top_n"]), fpn_post_nms_top_n=int(g["fpn_post_nms_top_n"])).eval()
xs = [scene(s, 300 + i).to(dev) for i, s in enumerate(g["shapes"])]
with torch.no_grad():
    boxes, _, scores = m(xs)
    # oracle on the same weights (CPU)
    from oracle import fcos as OF, nets as ON
    from fixture_init import seeded_state
    ob = ON.VGGFPN("EF", 4, 160); seeded_state(ob, 1)
    oh = OF.FCOSHead(256, 4, [4, 8, 16, 32], True, True, rot)
    oh.load_state_dict({k: v.cpu() for k, v in m.fcos_module.head.state_dict().items()})
    orc = OF.FCOS(ob, oh, [4, 8, 16, 32], rot, pre_nms_top_n=int(g["pre_nms_top_n"]), fpn_post_nms_top_n=int(g["fpn_post_nms_top_n"]))
    ob.eval()
    oboxes, _, oscores, aux = orc([x.cpu() for x in xs])
    feats = m.backbone(torch.stack(m.transform(list(xs))))
    lg, rg, ct = m.fcos_module.head(list(feats))
    for l in range(4):
        print("level", l, "cls", (lg[l].cpu() - aux["box_cls"][l]).abs().max().item(), "reg", (rg[l].cpu() - aux["box_reg"][l]).abs().max().item(),
              "ctr", (ct[l].cpu() - aux["centerness"][l]).abs().max().item())
for i in range(len(xs)):
    rp, rs = T(g[f"boxes{i}"]), T(g[f"scores{i}"])
    gp, gs = boxes[i].cpu(), scores[i].cpu()
    print("scene", i, gp.shape, rp.shape, oboxes[i].shape)
    near = (gs[None, :] - rs[:, None]).abs() <= 3e-6
    diff = (gp[None, :, 1:] - rp[:, None, 1:]).abs()
    ok = ((diff <= 3e-3 + 2e-4 * rp[:, 1:].abs()[:, None, :]).all(dim=2) & near).any(dim=1)
    ok2 = ((diff <= 3e-3 + 2e-4 * rp[:, 1:].abs()[:, None, :]).all(dim=2) & near).any(dim=0)
    print(" ref rows unmatched:", torch.where(~ok)[0].tolist())
    for r in torch.where(~ok)[0][:6]:
        print("   ref", r.item(), rp[r].tolist(), rs[r].item())
    print(" got rows unmatched:", torch.where(~ok2)[0].tolist())
    for r in torch.where(~ok2)[0][:6]:
        print("   got", r.item(), gp[r].tolist(), gs[r].item())

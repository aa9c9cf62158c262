import sys, os
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from test_gpu_e2e import build, scene, T
from fixture_init import seeded_state
from oracle import nets as ON, rpn as OR
dev = torch.device('cuda:0')
g = dict(np.load(os.path.join(R, 'tests', 'golden', 'train_aabb.npz')))
m = build(False, 160, dev).train()
xs = [scene(s, 200 + i) for i, s in enumerate(g['shapes'])]
gts = [T(g[f'gt{i}']) for i in range(len(xs))]
pos, neg = T(g['pos_idx']), T(g['neg_idx'])
m.rpn.sampler_hook = lambda labels: (pos.to(dev), neg.to(dev))
x = torch.stack([t.to(dev) for t in xs])
feats = list(m.backbone(x))
for f in feats: f.retain_grad()
_, _, losses, _ = m.rpn(x, feats, [tuple(xs[0].shape[-3:])], [t.to(dev) for t in gts])
(losses['loss_objectness'] + 5.0 * losses['loss_rpn_box_reg']).backward()
# oracle
obb, ohd = ON.VGGFPN('EF', 4, 160), ON.RPNHead(256, 13, 4, False)
seeded_state(obb, 1); seeded_state(ohd, 2); obb.train()
orpn = OR.RPN(ohd, rotated=False); orpn.sampler_hook = lambda labels: (pos, neg)
xo = torch.stack(xs)
of = list(obb(xo))
for f in of: f.retain_grad()
_, _, ol, _, aux = orpn(xo, of, [tuple(xs[0].shape[-3:])], gts, True)
(ol['loss_objectness'] + 5.0 * ol['loss_rpn_box_reg']).backward()
for i, (a, b) in enumerate(zip(feats, of)):
    ga, gb = a.grad.float().cpu(), b.grad
    print('level', i, tuple(gb.shape), 'feat err', (a.detach().float().cpu() - b.detach()).abs().max().item(), 'grad max', gb.abs().max().item(), 'grad err', (ga - gb).abs().max().item(),
          'nnz ref', int((gb != 0).sum()), 'nnz mine', int((ga != 0).sum()))
    d = (ga - gb).abs()
    idx = d.reshape(-1).argmax().item()
    print('   worst at', np.unravel_index(idx, tuple(gb.shape)), ga.reshape(-1)[idx].item(), gb.reshape(-1)[idx].item())
    print('   per-voxel err', d.amax(dim=1).reshape(-1)[:30].tolist())
ho = dict(ohd.named_parameters()); hm = dict(m.rpn.head.named_parameters())
for k in ho:
    print(k, 'grad err', (hm[k].grad.cpu() - ho[k].grad).abs().max().item(), 'scale', ho[k].grad.abs().max().item())
T_ = aux['anchors'].shape[0]
print('pos levels', np.histogram(pos.numpy(), bins=[0, 1728*13, (1728+216)*13, (1728+216+27)*13, T_])[0], 'neg levels', np.histogram(neg.numpy(), bins=[0, 1728*13, (1728+216)*13, (1728+216+27)*13, T_])[0])

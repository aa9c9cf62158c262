import sys, os
import numpy as np, torch
_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, _R)
sys.path.insert(0, os.path.join(_R, 'tests'))
from test_gpu_e2e import build, scene, T
dev = torch.device('cuda:0')
G = lambda n: dict(np.load(os.path.join(_R, 'tests', 'golden', n + '.npz')))
g = G('eval_aabb_s2')
m = build(False, 160, dev).eval()
with torch.no_grad():
    (feats, props, lvls), _, scores = m([scene(g['shapes'][0], 100).to(dev)])
for i, f in enumerate(feats):
    got = f.float().contiguous().reshape(-1)[T(g[f'feat{i}_idx'], dev)].cpu(); ref = T(g[f'feat{i}_val'])
    print('feat', i, 'max abs err', (got - ref).abs().max().item(), 'ref max', ref.abs().max().item())
for name in ('train_aabb', 'train_obb_giou'):
    g = G(name)
    m = build(bool(g['rotated']), 160, dev, str(g['reg_loss_type'])).train()
    xs = [scene(s, 200 + i).to(dev) for i, s in enumerate(g['shapes'])]
    gts = [T(g[f'gt{i}'], dev) for i in range(len(xs))]
    pos, neg = T(g['pos_idx'], dev), T(g['neg_idx'], dev)
    m.rpn.sampler_hook = lambda labels: (pos, neg)
    _, losses, _ = m(xs, gts)
    print(name, {k: (v.item(), float(g[k])) for k, v in losses.items()})
    (losses['loss_objectness'] + 5.0 * losses['loss_rpn_box_reg']).backward()
    params = dict(m.backbone.named_parameters()); params.update({'head.' + k: v for k, v in m.rpn.head.named_parameters()})
    rows = []
    for k, p in params.items():
        if 'grad/' + k in g: ref, got, r64 = T(g['grad/' + k]), p.grad.cpu(), T(g['grad64/' + k])
        else: ref, got, r64 = T(g['gval/' + k]), p.grad.reshape(-1)[T(g['gidx/' + k], dev)].cpu(), T(g['gval64/' + k])
        sc = float(g['gmax64/' + k])
        if sc < 1e-7: continue
        rows.append(((got - ref).abs().max().item() / sc, (got.double() - r64).abs().max().item() / sc, (ref.double() - r64).abs().max().item() / sc, k))
    rows.sort(reverse=True)
    print('  rel err: mine-vs-ref32 | mine-vs-f64 | ref32-vs-f64(subsample)')
    for r in rows[:14]: print('  %.2e %.2e %.2e %s' % r)

"""Why does a reference proposal row have no partner?  Runs the given eval fixtures and prints, for every row without an exact partner,
what the HIP path's own stage tensors say (tests/test_gpu_e2e.py::_explain_unmatched mechanisms + raw numbers)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_e2e import T, build, scene  # noqa: E402
from oracle import boxes as OB  # noqa: E402

dev = torch.device('cuda:0')
for name in sys.argv[1:]:
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'), allow_pickle=True))
    rot = bool(g['rotated'])
    m = build(rot, int(g['resolution']), dev, pre=int(g['pre']), backbone=str(g.get('backbone', 'vgg'))).eval()
    if 'shapes' in g:
        xs = [scene(s, 100 + i).to(dev) for i, s in enumerate(g['shapes'])]
    else:
        xs = [torch.rand(4, *[int(s) for s in g['shape']], generator=torch.Generator().manual_seed(int(g['seed']))).to(dev)]
    with torch.no_grad():
        (feats, props, lvls), _, scores = m(xs)
    size = torch.tensor([float(v) for v in xs[0].shape[1:]])
    rp, rs, rl = T(g['proposals0']), T(g['scores0']), T(g['levels0'])
    gp, gs, gl = props[0].cpu(), scores[0].cpu(), lvls[0].cpu()
    near = (gs[None, :] - rs[:, None]).abs() <= 2e-6
    diff = (gp[None] - rp[:, None]).abs()
    tol = 2e-3 + 1e-4 * rp.abs()[:, None, :]
    ok = ((diff <= tol).all(2) & near & (gl[None] == rl[:, None])).any(1)
    bad = torch.where(~ok)[0]
    st = m.rpn.last_aux['stages'][0]
    cb, cv, cl = st['cand_boxes'].cpu(), st['cand_valid'].cpu().bool(), st['cand_level'].cpu().long()
    print(f'== {name}: {bad.numel()} rows without a partner of {rp.shape[0]} (HIP has {gp.shape[0]})')
    for l in range(4):
        sel = (cl == l) & cv
        if sel.any() and rot:
            c = cb[sel][:, :3]
            dist = torch.minimum(c.abs(), (c - size).abs()).min(dim=1).values
            print(f'   level {l}: {int(sel.sum())} candidates; closest centre-to-face distance {dist.min().item():.3e}; outside: {int(((c < 0) | (c > size)).any(1).sum())}')
    for b in bad[:40].tolist():
        lvl = int(rl[b])
        same = gl.long() == lvl
        boxok = same & ((gp - rp[b]).abs() <= tol[b, 0]).all(1)
        j = torch.where(boxok)[0]
        iou = OB.iou_matrix(rp[b][None].double(), gp[same].double())[0] if rot else OB.aabb_iou_matrix(rp[b][None].double(), gp[same].double())[0]
        print(f'   row {b} lvl {lvl} score {rs[b]:.7f} box {[round(v, 4) for v in rp[b].tolist()]}')
        if j.numel():
            print(f'      same box at HIP row {int(j[0])} with score {gs[j[0]]:.7f} (diff {gs[j[0]] - rs[b]:+.2e})')
            if 'cand_logits' in st:      # slot analysis of the B3 mechanism (tests/test_gpu_e2e.py::_explain_unmatched)
                sel = (cl == lvl) & cv
                sc = torch.sigmoid(st['cand_logits'].float().cpu())[sel]
                c = cb[:, :3]
                ctol = 2e-3 + 1e-4 * c.abs()
                at_face = (((c.abs() <= ctol) | ((c - size).abs() <= ctol)).any(dim=1) & cv)[cl <= lvl].sum()      # cumulative: the clip shifts ALL later candidates
                ties = int(((sc[:-1] - sc[1:]).abs() <= 2e-6).sum())
                dr, dh = (sc - rs[b]).abs(), (sc - gs[j[0]]).abs()
                print(f'      level list: {sc.numel()} candidates, {int(at_face)} at a face, {ties} adjacent ties; reference score sits at slot {int(dr.argmin())} '
                      f'(|d| {dr.min().item():.1e}), HIP score at slot {int(dh.argmin())} (|d| {dh.min().item():.1e}); outside the grid: '
                      f'{int((((c < 0) | (c > size)).any(dim=1) & cv)[cl == lvl].sum())}')
        else:
            top = torch.topk(iou, min(3, iou.numel()))
            print(f'      no HIP box within tolerance; best IoUs with HIP proposals of the level: {[round(v, 5) for v in top.values.tolist()]}; '
                  f'min |IoU - 0.3| = {(iou - 0.3).abs().min().item():.2e}')

import sys, os
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from fixture_init import seeded_state
from oracle import nets as ON
from nerf_rpn_amd.model import RPNHead
from nerf_rpn_amd.model import hip_nn
dev = torch.device('cuda:0')
torch.manual_seed(0)
ohd = ON.RPNHead(256, 13, 4, False); seeded_state(ohd, 2)
hd = RPNHead(256, 13, 4, rotate=False); seeded_state(hd, 2); hd = hd.to(dev)
for grid in [(3, 3, 3), (6, 6, 6), (12, 12, 12)]:
    f = torch.randn(1, 256, *grid) * 0.5
    fo = f.clone().requires_grad_(True)
    lo, bo = ohd([fo])
    gl = torch.zeros_like(lo[0]); gb = torch.zeros_like(bo[0])
    gl[0, 3, 1, 1, 0] = 2e-3; gl[0, 7, 2, 0, 1] = 1.5e-3; gb[0, 10, 1, 2, 2] = -3e-3
    (lo[0] * gl).sum().backward(retain_graph=True); (bo[0] * gb).sum().backward()
    fm = f.clone().to(dev).requires_grad_(True)
    lm, bm = hd([fm])   # plain NCDHW input: converted by a (differentiable) kernel
    ((lm[0] * gl.to(dev)).sum() + (bm[0] * gb.to(dev)).sum()).backward()
    print(grid, 'fwd logits err', (lm[0].detach().cpu() - lo[0].detach()).abs().max().item(), 'dfeat max', fo.grad.abs().max().item(),
          'dfeat err', (fm.grad.cpu() - fo.grad).abs().max().item())
    for (k, a), (_, b) in zip(hd.named_parameters(), ohd.named_parameters()):
        print('   ', k, 'rel err', ((a.grad.cpu() - b.grad).abs().max() / (b.grad.abs().max() + 1e-30)).item())
    hd.zero_grad(); ohd.zero_grad()

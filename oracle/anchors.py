"""ORACLE (test infrastructure only): 3D anchor grid on CPU.

Restates reference nerf_rpn/model/anchor.py:14-174 with the constants of run_rpn.py:31-35.
"""
import itertools

import torch

SIZES = ((8,), (16,), (32,), (64,))
RATIOS = ((1., 1., 1.), (1., 1., 2.), (1., 2., 2.), (1., 1., 3.), (1., 3., 3.))

# anchor.py:57-60 iterates ``set(itertools.permutations(ratio))`` -- CPython hash order of float
# tuples (quirk B1).  This is the order observed under CPython 3.10; it fixes the channel <-> anchor
# mapping of released checkpoints.  make_golden.py asserts it against the live reference.
RATIO_ORDER = ((1., 1., 1.),
               (1., 2., 1.), (2., 1., 1.), (1., 1., 2.),
               (1., 2., 2.), (2., 1., 2.), (2., 2., 1.),
               (1., 1., 3.), (1., 3., 1.), (3., 1., 1.),
               (3., 1., 3.), (3., 3., 1.), (1., 3., 3.))


def base_anchors(scales, ratio_order=RATIO_ORDER):
    """anchor.py:49-82 (is_normalized=False): -> [A,6] rounded half-to-even."""
    r = torch.tensor(ratio_order, dtype=torch.float32)
    s = torch.as_tensor(scales, dtype=torch.float32)
    e = (r[:, None, :] * s[None, :, None]).reshape(-1, 3)
    return (torch.cat([-e, e], dim=1) / 2).round()


def level_anchors(grid, stride, base):
    """anchor.py:98-122: anchors ordered (x, y, z, a) -> [gx*gy*gz*A, 6]."""
    sx = torch.arange(grid[0], dtype=torch.float32) * stride[0]
    sy = torch.arange(grid[1], dtype=torch.float32) * stride[1]
    sz = torch.arange(grid[2], dtype=torch.float32) * stride[2]
    X, Y, Z = torch.meshgrid(sx, sy, sz, indexing="ij")
    sh = torch.stack([X, Y, Z, X, Y, Z], dim=-1).reshape(-1, 1, 6)
    return (sh + base.view(1, -1, 6)).reshape(-1, 6)


def all_anchors(mesh_size, grids, sizes=SIZES):
    """anchor.py:154-174: per-level list; stride = mesh // grid per axis (quirk B2)."""
    out = []
    for g, s in zip(grids, sizes):
        stride = [mesh_size[i] // g[i] for i in range(3)]
        out.append(level_anchors(g, stride, base_anchors(s)))
    return out


def padding_masks(mesh_size, grids, ori_sizes, num_anchors=13):
    """anchor.py:124-152 + rpn.py:230-238: bool [N, sum(g^3*A)] in (x,y,z,a) order."""
    per_level = []
    for g in grids:
        stride = torch.tensor([mesh_size[i] // g[i] for i in range(3)])
        rows = []
        for o in ori_sizes:
            lim = torch.ceil(torch.tensor(o) / stride).to(torch.int64)
            m = torch.zeros((g[0], g[1], g[2], num_anchors), dtype=torch.bool)
            m[:lim[0], :lim[1], :lim[2], :] = True
            rows.append(m.reshape(-1))
        per_level.append(torch.stack(rows))
    return torch.cat(per_level, dim=1)

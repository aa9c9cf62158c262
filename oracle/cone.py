"""Test infrastructure (CPU oracle): the sampled-anchor cone lists of the RPN head, restated in numpy / torch.

The RPN loss reads the head at the sampled anchors only (reference model/rpn.py:389-420; "During training, boxes pred and scores are
unused", rpn.py:506), so the last hidden map of the head (reference model/anchor.py RPNHead: conv_depth x [Conv3d k3 + ReLU], then the 1x1x1
cls / bbox convs) is needed on the voxels S0 that hold a sampled anchor, the map before on S1 = the 3x3x3 dilation of S0 inside its own
grid, and so on.  ``cone_lists`` returns those sets as the HIP kernels (csrc/cone.hip) emit them: ascending voxel ids of the ragged
(level-major, scene-major inside a level) voxel space and, per voxel, the tap word = in-bounds bits of its 27 neighbours | segment << 27.
Only tests/ may import this module."""
import numpy as np
import torch
import torch.nn.functional as F


def cone_lists(pos_per_scene, neg_per_scene, grids, num_anchors, depth):
    """pos/neg_per_scene: per scene, 1-D integer arrays of anchor indices in one scene's flat (level, x, y, z, a) order.
    grids: [(X, Y, Z)] per level.  -> [ (ids uint32 [n_k], words uint32 [n_k]) for k = 0..depth ]"""
    n = len(pos_per_scene)
    cells = [g[0] * g[1] * g[2] for g in grids]
    level_off = np.concatenate([[0], np.cumsum([c * num_anchors for c in cells])])
    masks = []            # per segment (level-major, scene-major): bool grid
    for l, g in enumerate(grids):
        for s in range(n):
            a = np.concatenate([np.asarray(pos_per_scene[s]).reshape(-1), np.asarray(neg_per_scene[s]).reshape(-1)]).astype(np.int64)
            a = a[(a >= level_off[l]) & (a < level_off[l + 1])] - level_off[l]
            m = np.zeros(cells[l], dtype=bool)
            m[np.unique(a // num_anchors)] = True
            masks.append(m.reshape(g))
    seg_grids = [g for g in grids for _ in range(n)]
    starts = np.concatenate([[0], np.cumsum([g[0] * g[1] * g[2] for g in seg_grids])])
    out = []
    cur = [torch.from_numpy(m.astype(np.float32))[None, None] for m in masks]
    for k in range(depth + 1):
        if k > 0:
            cur = [(F.max_pool3d(c, 3, 1, 1) > 0).float() for c in cur]
        ids, words = [], []
        for seg, (c, g) in enumerate(zip(cur, seg_grids)):
            loc = np.nonzero(c.reshape(-1).numpy() > 0)[0]
            X, Y, Z = g
            x, y, z = loc // (Y * Z), (loc // Z) % Y, loc % Z
            w = np.full(loc.shape, seg << 27, dtype=np.uint32)
            for t in range(27):
                dx, dy, dz = t // 9 - 1, (t // 3) % 3 - 1, t % 3 - 1
                inb = (x + dx >= 0) & (x + dx < X) & (y + dy >= 0) & (y + dy < Y) & (z + dz >= 0) & (z + dz < Z)
                w |= (inb.astype(np.uint32) << np.uint32(t))
            ids.append((loc + starts[seg]).astype(np.uint32))
            words.append(w)
        out.append((np.concatenate(ids), np.concatenate(words)))
    return out

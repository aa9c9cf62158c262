"""Small OBB helpers with the reference's names (coder/misc.py:3-101), HIP-backed where a kernel exists."""
import torch

from ... import ops

pi = 3.141592


def obb2hbb_3d(obboxes):
    return ops.obb_to_aabb(obboxes)


def obb2points_3d(obboxes):
    """[K,7] -> [2K,3] extreme points used by the 2-D projection loss (differentiable torch ops, K <= 128)."""
    center, w, l, h, theta = torch.split(obboxes, [3, 1, 1, 1, 1], dim=-1)
    c, s = torch.cos(theta), torch.sin(theta)
    v = torch.cat([w / 2 * c - l / 2 * s, w / 2 * s + l / 2 * c, h / 2], dim=-1)
    return torch.cat([center - v, center + v], dim=0)

"""FPN level assignment of RoIs (reference nerf_rpn/model/level_mapper.py:8-83, after torchvision's poolers): level =
floor(lvl0 + log2(cbrt(w*l*h) / s0) + eps), clamped to the pyramid, returned relative to the finest level."""
import torch
from torch import Tensor


def box_volume(boxes: Tensor) -> Tensor:
    return boxes[..., 3] * boxes[..., 4] * boxes[..., 5]


class LevelMapper:
    def __init__(self, k_min: int, k_max: int, canonical_scale: int = 160, canonical_level: int = 4, eps: float = 1e-6):
        self.k_min, self.k_max, self.s0, self.lvl0, self.eps = k_min, k_max, canonical_scale, canonical_level, eps

    def __call__(self, boxlists: Tensor) -> Tensor:
        s = torch.pow(box_volume(boxlists), 1.0 / 3.0)
        lvl = torch.floor(self.lvl0 + torch.log2(s / self.s0) + torch.tensor(self.eps, dtype=s.dtype))
        lvl = torch.clamp(lvl, min=self.k_min, max=self.k_max)
        return (lvl.to(torch.int64) - self.k_min).to(torch.int64)


def initLevelMapper(k_min, k_max, canonical_scale=160, canonical_level=4, eps=1e-6):
    return LevelMapper(k_min, k_max, canonical_scale, canonical_level, eps)


def _setup_scales(scales, canonical_scale, canonical_level):
    lvl_min = -torch.log2(torch.tensor(scales[0], dtype=torch.float32)).item()
    lvl_max = -torch.log2(torch.tensor(scales[-1], dtype=torch.float32)).item()
    return initLevelMapper(int(lvl_min), int(lvl_max), canonical_scale=canonical_scale, canonical_level=canonical_level)

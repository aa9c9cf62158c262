"""Scene loading with the reference's dataset class names and ``__getitem__`` contract
(reference nerf_rpn/datasets.py:14-330): ``(rgbsigma [4,W,L,H] float32, boxes [G,6|7] | None, scene_name)``.

Host-side I/O only (npz / npy / csv -> torch CPU tensors); the arithmetic of the hot path starts when the trainer moves the
grid to the GPU.  ``density_to_alpha`` keeps the reference formula alpha = clip(1 - exp(-exp(sigma) / 100), 0, 1)
(datasets.py:165-167; ScanNet variant with a ReLU activation, :227-231)."""
import os
import random
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


def density_to_alpha(density):
    return np.clip(1.0 - np.exp(-np.exp(density) / 100.0), 0.0, 1.0)


def density_to_alpha_relu(density):
    return np.clip(1.0 - np.exp(-np.clip(density, a_min=0, a_max=None) / 100.0), 0.0, 1.0)


class AugPlan:
    """One draw of the training augmentation (reference datasets.py:109-163): which of the 90-degree rotation / axis flips /
    rotate-and-scale apply to a scene.  The host path applies it with torch ops, the device path inside ``nrpn_ingest_augment``."""
    __slots__ = ("rot90", "flips", "angle", "scale", "z_up")

    def __init__(self, rot90=False, flips=(False, False), angle=None, scale=None, z_up=True):
        self.rot90, self.flips, self.angle, self.scale, self.z_up = rot90, tuple(flips), angle, scale, z_up

    @property
    def identity(self):
        return not (self.rot90 or any(self.flips) or self.angle is not None)

    def xform(self):
        """R(angle) * scale exactly as the reference builds it (float64 trig -> float32 tensor -> * scale), or None."""
        if self.angle is None:
            return None
        a = self.angle
        return torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float) * self.scale


class RawScene:
    """A scene still in its on-disk (W,L,H,4) layout (float32 or uint8), to be finished ON THE DEVICE: ``ops.ingest_rgbsigma``
    (uint8 -> /255, density_to_alpha, cast to the compute dtype, channels-last) or, with an augmentation plan,
    ``ops.ingest_augment`` (the same plus rotation / flips / rotate-and-scale resampling in the same pass) -- instead of numpy alpha +
    host transpose + fp32 conversion + torch flips + grid_sample on the host.  Produced by datasets built with ``device_ingest=True``."""
    __slots__ = ("data", "alpha_mode", "plan")

    def __init__(self, data, alpha_mode, plan=None):
        self.data, self.alpha_mode, self.plan = data, alpha_mode, plan

    @property
    def shape(self):            # the logical [4,W,L,H] shape AFTER augmentation, so len()/shape based bookkeeping keeps working
        w, l, h = (int(v) for v in self.data.shape[:3])
        if self.plan is not None and self.plan.rot90:
            w, l, h = (l, w, h) if self.plan.z_up else (h, l, w)
        return torch.Size((4, w, l, h))

    def pin_memory(self):
        self.data = self.data.pin_memory()
        return self

    def to_device(self, dtype=torch.float32):
        from . import ops
        raw = self.data.cuda(non_blocking=True)
        if self.plan is None or self.plan.identity:
            return ops.ingest_rgbsigma(raw, self.alpha_mode, dtype)
        return ops.ingest_augment(raw, self.alpha_mode, dtype, self.plan)


def _grid_from_npz(path, normalize_density, raw=False):
    with np.load(path) as f:
        g = f["rgbsigma"]
        # device ingest takes float32 grids, and uint8 grids without density_to_alpha (uint8 + alpha follows numpy's casts);
        # every other dtype the reference's host path accepts (float16 / float64 npz) goes through the host path below
        if raw and (g.dtype == np.float32 or (g.dtype == np.uint8 and not normalize_density)):
            return RawScene(torch.from_numpy(np.ascontiguousarray(g)), 1 if normalize_density else 0)
        if normalize_density:
            g[..., -1] = density_to_alpha(g[..., -1])
        t = torch.from_numpy(np.transpose(g, (3, 0, 1, 2)))          # (W,L,H,C) -> (C,W,L,H)
        if t.dtype == torch.uint8:
            t = t.float() / 255.0
    return t


def rotate_and_scale_scene(rgbsigma, boxes, angle, scale):
    """In-plane rotation by ``angle`` and isotropic scaling of a scene and its OBBs (datasets.py:291-329)."""
    assert boxes is None or boxes.shape[1] == 7
    xform = torch.tensor([[np.cos(angle), -np.sin(angle), 0], [np.sin(angle), np.cos(angle), 0], [0, 0, 1]], dtype=torch.float) * scale
    res = rgbsigma.shape[1:]
    axes = [torch.linspace(-1, 1, r) * r / 2 for r in res]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3) @ xform.T
    grid = grid[..., [2, 1, 0]].reshape(res[0], res[1], res[2], 3)
    for k, r in enumerate((res[2], res[1], res[0])):
        grid[..., k] = grid[..., k] / (r / 2)
    rgbsigma = F.grid_sample(rgbsigma.unsqueeze(0), grid.unsqueeze(0), align_corners=True).squeeze(0)
    if boxes is not None:
        boxes = boxes.clone()
        boxes[:, 6] = boxes[:, 6] - angle
        boxes[:, 3:6] = boxes[:, 3:6] / scale
        center = torch.tensor(res).unsqueeze(0) / 2
        boxes[:, :3] = (boxes[:, :3] - center) @ (xform.to(boxes.dtype) / (scale * scale)) + center
    return rgbsigma, boxes


class BaseDataset(torch.utils.data.Dataset):
    def __init__(self, dataset_type: str = None, features_path: str = None, boxes_path: str = None,
                 scene_list: Optional[List[str]] = None, normalize_density: bool = True, flip_prob: float = 0.0,
                 rotate_prob: float = 0.0, rot_scale_prob: float = 0.0, z_up: bool = True) -> None:
        super().__init__()
        self.dataset_type = dataset_type
        self.features_path, self.boxes_path = features_path, boxes_path
        self.scene_list = scene_list
        self.normalize_density = normalize_density
        self.flip_prob, self.rotate_prob, self.rot_scale_prob = flip_prob, rotate_prob, rot_scale_prob
        self.z_up = z_up
        self.scene_data = []
        self.device_ingest = False       # True: un-augmented scenes are returned as RawScene (finished on the GPU)

    density_to_alpha = staticmethod(density_to_alpha)

    def load_single_scene(self, scene: str):
        boxes = None if self.boxes_path is None else torch.from_numpy(np.load(os.path.join(self.boxes_path, scene + ".npy")))
        return scene, _grid_from_npz(os.path.join(self.features_path, scene + ".npz"), self.normalize_density, self.device_ingest), boxes

    def load_scene_data(self, preload: bool = False):
        if self.scene_list is None:
            self.scene_list = [f.split(".")[0] for f in os.listdir(self.features_path) if f.endswith(".npz")]
        kept = []
        for scene in self.scene_list:
            if not os.path.isfile(os.path.join(self.features_path, scene + ".npz")):
                print(f"{scene} does not have a feature file")
                continue
            if self.boxes_path is not None and np.load(os.path.join(self.boxes_path, scene + ".npy")).shape[0] == 0:
                print(f"{scene} does not have any boxes")
                continue
            kept.append(scene)
        self.scene_list = kept
        if preload:
            self.scene_data = [self.load_single_scene(s) for s in self.scene_list]

    def __getitem__(self, index: int):
        if self.scene_data:
            scene, rgbsigma, boxes = self.scene_data[index]
        else:
            scene = self.scene_list[index]
            _, rgbsigma, boxes = self.load_single_scene(scene)
        if self.flip_prob > 0 or self.rotate_prob > 0 or self.rot_scale_prob > 0:
            if isinstance(rgbsigma, RawScene):      # device ingest: draw the plan + move the boxes here, the voxels move on the GPU
                plan, boxes = self.draw_augmentation(rgbsigma.shape[1:], boxes, self.flip_prob, self.rotate_prob, self.rot_scale_prob, self.z_up)
                rgbsigma = RawScene(rgbsigma.data, rgbsigma.alpha_mode, plan)
            else:
                rgbsigma, boxes = self.augment_rpn_inputs(rgbsigma, boxes, self.flip_prob, self.rotate_prob, self.rot_scale_prob, self.z_up)
        return rgbsigma, boxes, scene

    def __len__(self) -> int:
        return len(self.scene_list)

    @staticmethod
    def draw_augmentation(shape, boxes, flip_prob: float, rotate_prob: float, rot_scale_prob: float, z_up: bool = True):
        """Consume python's ``random`` exactly like the reference's augment_rpn_inputs (datasets.py:109-163) and return the drawn
        ``AugPlan`` together with the transformed boxes; ``shape`` = (W, L, H) of the scene before augmentation."""
        for name, p in (("flip_prob", flip_prob), ("rotate_prob", rotate_prob), ("rotate_and_scale_prob", rot_scale_prob)):
            if p < 0 or p > 1:
                raise ValueError(f"{name} must be between 0 and 1, but got {p}")
        if boxes is not None:
            assert (z_up and boxes.shape[1] == 7) or boxes.shape[1] == 6, "z_up must be True when boxes are in (x, y, z, w, l, h, t) format"
        size = [int(v) for v in shape]
        plan = AugPlan(z_up=z_up)
        if random.random() < rotate_prob:
            plan.rot90 = True
            size = [size[1], size[0], size[2]] if z_up else [size[2], size[1], size[0]]
            if boxes is not None:
                boxes = boxes.clone()
                if boxes.shape[1] == 6:
                    if z_up:
                        boxes[:, [0, 1, 3, 4]] = boxes[:, [1, 0, 4, 3]]
                        boxes[:, [0, 3]] = size[0] - boxes[:, [3, 0]]
                    else:
                        boxes[:, [0, 2, 3, 5]] = boxes[:, [2, 0, 5, 3]]
                        boxes[:, [2, 5]] = size[2] - boxes[:, [5, 2]]
                else:
                    boxes[:, [0, 1, 3, 4]] = boxes[:, [1, 0, 4, 3]]
                    boxes[:, 0] = size[0] - boxes[:, 0]
        flips = [False, False]
        for n, axis in enumerate([0, 1] if z_up else [0, 2]):
            if random.random() < flip_prob:
                flips[n] = True
                if boxes is not None:
                    boxes = boxes.clone()
                    if boxes.shape[1] == 6:
                        boxes[:, [axis, axis + 3]] = size[axis] - boxes[:, [axis + 3, axis]]
                    else:
                        boxes[:, axis] = size[axis] - boxes[:, axis]
                        boxes[:, -1] = -boxes[:, -1]
        plan.flips = tuple(flips)
        if boxes is not None and boxes.shape[1] == 7 and random.random() < rot_scale_prob:
            plan.angle, plan.scale = random.uniform(-np.pi / 18, np.pi / 18), random.uniform(0.9, 1.1)
            xform = plan.xform()
            boxes = boxes.clone()
            boxes[:, 6] = boxes[:, 6] - plan.angle
            boxes[:, 3:6] = boxes[:, 3:6] / plan.scale
            center = torch.tensor(size).unsqueeze(0) / 2
            boxes[:, :3] = (boxes[:, :3] - center) @ (xform.to(boxes.dtype) / (plan.scale * plan.scale)) + center
        return plan, boxes

    @staticmethod
    def apply_plan_host(rgbsigma: Tensor, plan: "AugPlan") -> Tensor:
        """The voxel side of an AugPlan with torch ops on the host (the reference's own operations)."""
        if plan.rot90:
            a, b = (1, 2) if plan.z_up else (1, 3)
            rgbsigma = torch.flip(torch.transpose(rgbsigma, a, b), [a if plan.z_up else 3])
        for on, axis in zip(plan.flips, [0, 1] if plan.z_up else [0, 2]):
            if on:
                rgbsigma = rgbsigma.flip(dims=[axis + 1])
        if plan.angle is not None:
            rgbsigma, _ = rotate_and_scale_scene(rgbsigma, None, plan.angle, plan.scale)
        return rgbsigma

    @staticmethod
    def augment_rpn_inputs(rgbsigma: Tensor, boxes: Tensor, flip_prob: float, rotate_prob: float, rot_scale_prob: float,
                           z_up: bool = True) -> Tuple[Tensor, Tensor]:
        """90-degree rotation, axis flips, small rotation+scale (datasets.py:109-163); consumes python's ``random``."""
        plan, boxes = BaseDataset.draw_augmentation(rgbsigma.shape[1:], boxes, flip_prob, rotate_prob, rot_scale_prob, z_up)
        return BaseDataset.apply_plan_host(rgbsigma, plan), boxes

    @staticmethod
    def collate_fn(batch):
        return [b[0] for b in batch], [b[1] for b in batch], [b[2] for b in batch]


class Front3DRPNDataset(BaseDataset):
    def __init__(self, features_path, boxes_path, scene_list=None, normalize_density=True, flip_prob=0.0, rotate_prob=0.0,
                 rot_scale_prob=0.0, preload=False):
        super().__init__("3dfront", features_path, boxes_path, scene_list, normalize_density, flip_prob, rotate_prob, rot_scale_prob)
        self.load_scene_data(preload=preload)


class HypersimRPNDataset(BaseDataset):
    def __init__(self, features_path, boxes_path, scene_list=None, normalize_density=True, flip_prob=0.0, rotate_prob=0.0,
                 rot_scale_prob=0.0, preload=False):
        super().__init__("hypersim", features_path, boxes_path, scene_list, normalize_density, flip_prob, rotate_prob, rot_scale_prob)
        self.load_scene_data(preload=preload)


class ScanNetRPNDataset(BaseDataset):
    def __init__(self, scene_list, features_path, boxes_path, flip_prob=0.0, rotate_prob=0.0, rot_scale_prob=0.0):
        super().__init__("hypersim", features_path, boxes_path, scene_list, False, flip_prob, rotate_prob, rot_scale_prob, z_up=True)
        self.load_scene_data(preload=True)
        for scene in self.scene_data:
            g = scene[1]
            g[-1, ...] = torch.as_tensor(density_to_alpha_relu(g[-1].numpy()))

    density_to_alpha = staticmethod(density_to_alpha_relu)


class GeneralRPNDataset(BaseDataset):
    def __init__(self, csv_path, normalize_density: bool = True) -> None:
        super().__init__("general")
        import pandas as pd
        self.df = pd.read_csv(csv_path, dtype=str)
        self.normalize_density = normalize_density
        self.scene_list = []
        for row in self.df.itertuples():
            self.scene_list.append(row.scene)
            assert os.path.isfile(row.rgbsigma_path), f"{row.rgbsigma_path} does not exist"
            boxes = None
            if row.boxes_path != "None":
                assert os.path.isfile(row.boxes_path), f"{row.boxes_path} does not exist"
                boxes = torch.from_numpy(np.load(row.boxes_path))
            self.scene_data.append((row.scene, _grid_from_npz(row.rgbsigma_path, normalize_density), boxes))


class RPNClassificationDataset(torch.utils.data.Dataset):
    """Scenes for the second-stage network (reference datasets.py:330-424): per scene either the pre-extracted pyramid features
    (``level_features`` + ``resolution`` in the feature .npz) or, with ``fine_tune``, the raw rgb-sigma grid; the ground-truth boxes; and the
    first stage's proposals as RoI rows (level index, box) read from ``roi_path/<scene>.npz`` (keys ``level_indices``, ``proposals``)."""

    def __init__(self, features_path: str, boxes_path: str, roi_path: str, scene_names: Optional[List[str]] = None, fine_tune: bool = False,
                 normalize_density: bool = True, flip_prob: float = 0.0, rotate_prob: float = 0.0, rotate_scale_prob: float = 0.0):
        self.features_path, self.boxes_path, self.fine_tune = features_path, boxes_path, fine_tune
        self.flip_prob, self.rotate_prob, self.rotate_scale_prob = flip_prob, rotate_prob, rotate_scale_prob
        if scene_names is None:
            scene_names = [f.split(".")[0] for f in os.listdir(features_path) if f.endswith(".npz")]
        self.scene_data = []
        for name in scene_names:
            name = str(name)
            if not os.path.isfile(os.path.join(boxes_path, name + ".npy")) or not os.path.isfile(os.path.join(roi_path, name + ".npz")):
                print(f"{name} does not have a training file")
                continue
            with np.load(os.path.join(features_path, name + ".npz"), allow_pickle=True) as f:
                resolution = f["resolution"]
                if not fine_tune:
                    lf = f["level_features"]
                    feats = [torch.from_numpy(np.asarray(lf[i]).reshape(resolution[i]).astype(np.float32)) for i in range(len(lf))]
                else:
                    g = f["rgbsigma"].astype(np.float32)
                    if normalize_density:
                        g[..., -1] = density_to_alpha(g[..., -1])
                    feats = [torch.from_numpy(np.transpose(g, (3, 0, 1, 2)))]
            boxes = torch.from_numpy(np.load(os.path.join(boxes_path, name + ".npy")))
            with np.load(os.path.join(roi_path, name + ".npz"), allow_pickle=True) as fr:
                level_indices, proposals = fr["level_indices"], fr["proposals"]
            if fine_tune:       # drop proposals covering more than half of the scene (reference :386-395)
                keep = proposals[:, 3] * proposals[:, 4] * proposals[:, 5] / (resolution[0] * resolution[1] * resolution[2]) <= 0.5
                level_indices, proposals = level_indices[keep], proposals[keep]
            rois = torch.from_numpy(np.concatenate([level_indices[..., None], proposals], axis=1))
            self.scene_data.append((name, feats, boxes, rois))

    density_to_alpha = staticmethod(density_to_alpha)
    augment_rpn_inputs = staticmethod(BaseDataset.augment_rpn_inputs)

    def __len__(self) -> int:
        return len(self.scene_data)

    def __getitem__(self, index: int):
        name, feats, boxes, rois = self.scene_data[index]
        if self.fine_tune and (self.flip_prob > 0 or self.rotate_prob > 0 or self.rotate_scale_prob > 0):
            level_indices, n_gt = rois[..., :1], boxes.size(0)
            grid, moved = self.augment_rpn_inputs(feats[0], torch.cat([boxes, rois[..., 1:]]), self.flip_prob, self.rotate_prob,
                                                  self.rotate_scale_prob)
            boxes, rois, feats = moved[:n_gt], torch.cat([level_indices, moved[n_gt:]], dim=-1), [grid]
        return feats, boxes, rois, name

    @staticmethod
    def collate_fn(batch):
        return [b[0] for b in batch], [b[1] for b in batch], [b[2] for b in batch], [b[3] for b in batch]

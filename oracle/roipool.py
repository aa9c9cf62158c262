"""ROIPool without the RoIAlign op -- the reference CLI's default second-stage pooling -- restated in plain torch (CPU or any device).
Test infrastructure ONLY (imported by tests/): the product path is csrc/roipool.hip through nerf_rpn_amd.model.detector.ROIPool.

Follows nerf_rpn/model/detector.py of the reference: ``_adaptive_max_pool`` :377-384 / :425-432, ``normal_forward`` :397-438 (AABB integer crops),
the rotated 8-corner gather :264-395 with 'pooling' / 'interpolation'.  Pinned: bit-exact against outputs of the reference itself
(tests/golden/roipool.npz, tests/golden/make_golden.py::gen_roipool; tests/test_harness_cpu.py::test_roipool_torch_paths_match_reference)."""
import torch
import torch.nn.functional as F


class ROIPoolOracle:
    def __init__(self, output_size, spatial_scale, enlarge_scale=0.2, is_rotated_bbox=False, feature_extracting_type="pooling"):
        self.output_size = [int(v) for v in output_size]
        self.spatial_scale = list(spatial_scale)
        self.enlarge_scale = enlarge_scale
        self.is_rotated_bbox = is_rotated_bbox
        self.feature_extracting_type = feature_extracting_type

    def __call__(self, features, rois):
        return self._rotated_forward(features, rois) if self.is_rotated_bbox else self._normal_forward(features, rois)

    def _adaptive_max_pool(self, feat):
        """[C, a, b, c] -> [C, *output_size]: zero padding at the high side up to a multiple of the output size, then a max-pool whose kernel
        = stride = ceil(extent / output) (reference :377-384, :425-432)."""
        size = torch.tensor(feat.shape[1:], dtype=torch.float32)
        out = torch.tensor(self.output_size, dtype=torch.float32)
        kernel = torch.ceil(size / out).int()
        pad = (kernel * out.int() - size.int()).int()
        feat = F.pad(feat, (0, int(pad[2]), 0, int(pad[1]), 0, int(pad[0])))
        k = [int(v) for v in kernel]
        return F.max_pool3d(feat[None], kernel_size=k, stride=k)[0]

    def _normal_forward(self, features, rois):
        """AABB RoIs: the enlarged box in feature voxels, floor of both corners, the inclusive integer crop, adaptive max-pool (:397-438)."""
        out = []
        for f, r in zip(features, rois):
            r = r.reshape(-1, r.shape[-1])
            lv = r[..., 0].long()
            scale = torch.tensor(self.spatial_scale, dtype=r.dtype, device=r.device)[lv][:, None]
            roi = r[..., 1:]
            # the reference's AABB "enlargement" places the corners at centre -+ 0.5 * (half extent * (1 + enlarge)): the pooled box is
            # (1 + enlarge) / 2 of the RoI, not larger than it (:203-209).  Kept: it is what its RCNN weights were trained on.
            ext = (roi[..., 3:] - roi[..., :3]) / 2 * (1 + self.enlarge_scale)
            ctr = (roi[..., 3:] + roi[..., :3]) / 2
            pos = torch.floor(torch.cat([ctr - 0.5 * ext, ctr + 0.5 * ext], dim=-1) / scale)
            pos = pos.long().tolist()
            feats = []
            for j, p in enumerate(pos):
                crop = f[int(lv[j])][..., p[0]:p[3] + 1, p[1]:p[4] + 1, p[2]:p[5] + 1]
                feats.append(self._adaptive_max_pool(crop.float()))
            out.append(torch.stack(feats) if feats else f[0].new_zeros((0, f[0].shape[0], *self.output_size)))
        return out

    def _rotated_forward(self, features, rois):
        """OBB RoIs without the op (:264-395): a regular grid of ceil(extent / scale) points per RoI, rotated by theta about the box centre,
        each point the reference's 8-corner blend  sum_corners feat[corner] * (1 - |dx| |dy| |dz|) / 8  (not a trilinear interpolation;
        kept as it is), zero outside the map; then adaptive max-pool ('pooling') or a trilinear resize ('interpolation').  Like the
        reference, the RoI extents are enlarged IN PLACE (its ``enlarge_roi`` writes through the view it is given), so the caller's
        RoIs -- and the proposals decoded from them afterwards -- are the enlarged ones."""
        out = []
        fns = [(a, b, c) for a in (torch.floor, torch.ceil) for b in (torch.floor, torch.ceil) for c in (torch.floor, torch.ceil)]
        for f, r in zip(features, rois):
            flat = r.reshape(-1, r.shape[-1])              # a view: the in-place enlargement below reaches the caller's tensor
            lv = flat[..., 0].long()
            flat[..., 4:7] = flat[..., 4:7] * (1 + self.enlarge_scale)
            boxes = flat[..., 1:]
            pooled = [None] * flat.shape[0]
            for level in range(len(f)):
                sel = torch.nonzero(lv == level).view(-1)
                if sel.numel() == 0:
                    continue
                fm = f[level].float()
                dims = fm.shape
                lr = boxes[sel].float()
                sc = float(self.spatial_scale[level])
                gsz = torch.ceil(lr[:, 3:6] / sc).long().clamp_min(1)
                mx = [int(v) for v in gsz.max(dim=0).values]
                grid = torch.stack(torch.meshgrid(*[torch.arange(m, device=lr.device) for m in mx], indexing="ij"), dim=0).reshape(3, -1).float()
                pos = grid[None].repeat(lr.shape[0], 1, 1) - (gsz[..., None].float() - 1) / 2.0
                th = lr[:, 6]
                zero, one = torch.zeros_like(th), torch.ones_like(th)
                rot = torch.stack([torch.stack([torch.cos(th), -torch.sin(th), zero], dim=1),
                                   torch.stack([torch.sin(th), torch.cos(th), zero], dim=1),
                                   torch.stack([zero, zero, one], dim=1)], dim=1)
                pos = rot @ pos + lr[:, :3, None] / sc                                          # [n, 3, G]
                p = pos.permute(1, 0, 2).reshape(3, -1)
                inside = ((p[0] >= 0) & (p[0] <= dims[1] - 1) & (p[1] >= 0) & (p[1] <= dims[2] - 1) & (p[2] >= 0) & (p[2] <= dims[3] - 1))
                acc = 0.
                for fa, fb, fc in fns:
                    q = [fa(p[0]), fb(p[1]), fc(p[2])]
                    idx = [q[d].clamp(0, dims[d + 1] - 1).long() for d in range(3)]
                    w = (p[0] - q[0]).abs() * (p[1] - q[1]).abs() * (p[2] - q[2]).abs()
                    acc = acc + fm[:, idx[0], idx[1], idx[2]] * (1. - w[None])
                acc = acc * inside[None] / 8
                acc = acc.reshape(dims[0], lr.shape[0], *mx).permute(1, 0, 2, 3, 4)
                for k, j in enumerate(sel.tolist()):
                    g = [int(v) for v in gsz[k]]
                    crop = acc[k][:, :g[0], :g[1], :g[2]]
                    if self.feature_extracting_type == "pooling":
                        pooled[j] = self._adaptive_max_pool(crop)
                    else:
                        pooled[j] = F.interpolate(crop[None], size=tuple(self.output_size), mode="trilinear", align_corners=True)[0]
            out.append(torch.stack(pooled) if pooled else f[0].new_zeros((0, f[0].shape[0], *self.output_size)))
        return out



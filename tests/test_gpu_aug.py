"""GPU parity of the fused ingest + augmentation kernel (nrpn_ingest_augment) against the host loader's torch ops, which restate the
reference's augment_rpn_inputs / rotate_and_scale_scene (datasets.py:109-163, 291-329; pinned by tests/golden/datasets.npz in
test_harness_cpu.py): 90-degree rotation and flips are index remaps (exact), rotate-and-scale is a trilinear resample (1e-4)."""
import itertools
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_scene(g, normalize):
    from nerf_rpn_amd.datasets import density_to_alpha
    g = g.copy()
    if normalize:
        g[..., -1] = density_to_alpha(g[..., -1])
    t = torch.from_numpy(np.transpose(g, (3, 0, 1, 2)))
    return t.float() / 255.0 if t.dtype == torch.uint8 else t


@pytest.mark.parametrize("z_up", [True, False])
def test_rotation_and_flips_are_exact_index_remaps(z_up, dev):
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.datasets import AugPlan, BaseDataset
    rng = np.random.default_rng(0)
    g32 = (rng.random((14, 11, 9, 4), dtype=np.float32) * 8 - 4)
    g8 = rng.integers(0, 256, (7, 12, 10, 4), dtype=np.uint8)
    for g, normalize in ((g32, True), (g32, False), (g8, False)):
        host = _host_scene(g, normalize)
        for rot, f0, f1 in itertools.product([False, True], repeat=3):
            plan = AugPlan(rot90=rot, flips=(f0, f1), z_up=z_up)
            ref = BaseDataset.apply_plan_host(host, plan)
            got = ops.ingest_augment(torch.from_numpy(g).to(dev), 1 if normalize else 0, torch.float32, plan)
            assert got.shape == ref.shape and got.permute(1, 2, 3, 0).is_contiguous()
            assert torch.allclose(got.cpu(), ref, atol=2e-6, rtol=1e-6), (z_up, rot, f0, f1, (got.cpu() - ref).abs().max())


@pytest.mark.parametrize("angle,scale", [(0.15, 1.07), (-0.17, 0.91), (0.0, 1.0), (0.05, 1.1)])
def test_rotate_and_scale_resample(angle, scale, dev):
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.datasets import AugPlan, BaseDataset
    rng = np.random.default_rng(1)
    g = (rng.random((22, 18, 13, 4), dtype=np.float32) * 8 - 4)
    host = _host_scene(g, True)
    for rot, f0 in ((False, False), (True, True)):
        plan = AugPlan(rot90=rot, flips=(f0, False), angle=angle, scale=scale, z_up=True)
        ref = BaseDataset.apply_plan_host(host, plan)
        got = ops.ingest_augment(torch.from_numpy(g).to(dev), 1, torch.float32, plan).cpu()
        assert got.shape == ref.shape
        err = (got - ref).abs()
        # fp32 sampling positions differ in the last bits between the host's linspace/matmul and the kernel's index arithmetic: a
        # position within ~1e-5 of a voxel boundary moves one tap in or out of the zero padding, everything else agrees to 1e-4
        assert (err > 1e-4).float().mean().item() < 2e-4, (err.max(), (err > 1e-4).float().mean())
        assert err.max() < 2e-2
    bf = ops.ingest_augment(torch.from_numpy(g).to(dev), 1, torch.bfloat16, AugPlan(angle=angle, scale=scale)).float().cpu()
    assert (bf - BaseDataset.apply_plan_host(host, AugPlan(angle=angle, scale=scale))).abs().max() < 2e-2


def test_dataset_device_path_equals_host_path(tmp_path, dev):
    """Same python ``random`` stream -> the dataset's device path (RawScene + AugPlan finished by the kernel) and its host path give
    the same boxes and the same voxels."""
    from nerf_rpn_amd.datasets import Front3DRPNDataset, RawScene
    rng = np.random.default_rng(5)
    (tmp_path / "f").mkdir(); (tmp_path / "b").mkdir()
    for name in ("s0", "s1", "s2"):
        np.savez(tmp_path / "f" / f"{name}.npz", rgbsigma=(rng.random((20, 16, 12, 4), dtype=np.float32) * 6 - 3))
        ctr = rng.random((5, 3)) * 8 + 4
        np.save(tmp_path / "b" / f"{name}.npy", np.concatenate([ctr, rng.random((5, 3)) * 4 + 1, rng.random((5, 1)) - 0.5], axis=1).astype(np.float32))
    kw = dict(features_path=str(tmp_path / "f"), boxes_path=str(tmp_path / "b"), scene_list=["s0", "s1", "s2"], normalize_density=True,
              flip_prob=0.5, rotate_prob=0.5, rot_scale_prob=0.6)
    host_ds, dev_ds = Front3DRPNDataset(**kw), Front3DRPNDataset(**kw)
    dev_ds.device_ingest = True
    seen_aug = 0
    for rnd in range(4):
        for i in range(3):
            random.seed(100 * rnd + i)
            hx, hb, _ = host_ds[i]
            random.seed(100 * rnd + i)
            dx, db, _ = dev_ds[i]
            assert isinstance(dx, RawScene) and tuple(dx.shape) == tuple(hx.shape)
            assert torch.allclose(db, hb, atol=1e-5)
            got = dx.to_device(torch.float32).cpu()
            err = (got - hx).abs()
            assert (err > 1e-4).float().mean().item() < 2e-4 and err.max() < 2e-2
            seen_aug += int(not dx.plan.identity)
    assert seen_aug >= 6

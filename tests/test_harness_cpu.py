"""CPU: the harness around the hot path -- command-line flags identical to the reference's, dataset loaders returning what
the reference's return on the same files, oracle metrics against golden values."""
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_cli_flags_match_reference():
    from nerf_rpn_amd.run_rpn import build_parser
    ref = json.load(open(os.path.join(GOLDEN, "cli_flags.json")))
    mine = {a.dest: a for a in build_parser()._actions if a.option_strings and a.dest != "help"}
    for f in ref:
        a = mine.pop(f["dest"])
        assert a.option_strings == f["options"], f["dest"]
        assert a.default == f["default"], (f["dest"], a.default, f["default"])
        assert (list(a.choices) if a.choices else None) == f["choices"], f["dest"]
        assert (a.type.__name__ if a.type else None) == f["type"], f["dest"]
        assert type(a).__name__ == f["action"], f["dest"]
    assert sorted(mine) == ["dtype", "fix_obb_clip"]          # the only additions
    from nerf_rpn_amd.run_rpn import parse_gpu_ids
    assert parse_gpu_ids("0-3") == [0, 1, 2, 3] and parse_gpu_ids("0,2,5-6") == [0, 2, 5, 6] and parse_gpu_ids("") == []


def test_datasets_match_reference(tmp_path, golden):
    from nerf_rpn_amd import datasets as D
    g = golden("datasets")
    os.makedirs(tmp_path / "f"); os.makedirs(tmp_path / "b")
    np.savez(tmp_path / "f" / "s0.npz", rgbsigma=g["g32"]); np.savez(tmp_path / "f" / "s1.npz", rgbsigma=g["g8"])
    np.save(tmp_path / "b" / "s0.npy", g["boxes"]); np.save(tmp_path / "b" / "s1.npy", g["boxes"][:1])
    ds = D.Front3DRPNDataset(str(tmp_path / "f"), str(tmp_path / "b"), scene_list=["s0", "s1"], normalize_density=True)
    x0, b0, n0 = ds[0]
    x1, b1, n1 = ds[1]
    assert n0 == "s0" and tuple(x0.shape) == (4, 12, 10, 8) and x0.dtype == torch.float32
    assert torch.equal(x0, torch.from_numpy(g["x0"])) and torch.equal(x1, torch.from_numpy(g["x1"]))
    random.seed(5)
    xa, ba = D.BaseDataset.augment_rpn_inputs(x0.clone(), b0.clone(), 1.0, 1.0, 1.0, True)
    assert torch.allclose(xa, torch.from_numpy(g["aug_x"]), atol=1e-6) and torch.allclose(ba, torch.from_numpy(g["aug_boxes"]), atol=1e-6)
    assert np.allclose(D.ScanNetRPNDataset.density_to_alpha(np.linspace(-3, 400, 9)), g["scannet_alpha"])
    rs, bs, ns = D.BaseDataset.collate_fn([ds[0], ds[1]])
    assert len(rs) == 2 and ns == ["s0", "s1"]


def test_oracle_metrics(golden):
    from oracle import metrics as OM
    g = golden("metrics")
    for tag in ("aabb", "obb"):
        P = [torch.from_numpy(g[f"{tag}_props{i}"]) for i in range(3)]
        S = [torch.from_numpy(g[f"{tag}_scores{i}"]) for i in range(3)]
        G = [torch.from_numpy(g[f"{tag}_gt{i}"]) for i in range(3)]
        assert abs(OM.recall(P, S, G, torch.tensor([0.5]), 40)["ar"].item() - float(g[f"{tag}_r50"])) < 1e-6
        assert abs(OM.recall(P, S, G, torch.arange(0.25, 1.0, 0.05), 100)["ar"].item() - float(g[f"{tag}_ar"])) < 1e-6
        assert abs(OM.average_precision(P, S, G, 0.5)["ap"].item() - float(g[f"{tag}_ap50"])) < 1e-6
        assert abs(OM.average_precision(P, S, G, 0.25, 50)["ap"].item() - float(g[f"{tag}_ap25"])) < 1e-6


def test_detect_cli_flags_match_reference():
    from nerf_rpn_amd.run_rpn_detect import build_parser
    ref = json.load(open(os.path.join(GOLDEN, "cli_flags_detect.json")))
    mine = {a.dest: a for a in build_parser()._actions if a.option_strings and a.dest != "help"}
    for f in ref:
        a = mine.pop(f["dest"])
        assert a.option_strings == f["options"], f["dest"]
        assert a.default == f["default"], (f["dest"], a.default, f["default"])
        assert (list(a.choices) if a.choices else None) == f["choices"], f["dest"]
        assert (a.type.__name__ if a.type else None) == f["type"], f["dest"]
        assert type(a).__name__ == f["action"] and a.nargs == f["nargs"], f["dest"]
    assert not mine


def test_rotated_coder_and_level_mapper_match_reference(golden):
    """The second stage's CPU-side arithmetic against what the reference produced (tests/golden/make_golden.py::gen_detector)."""
    from nerf_rpn_amd.model.coder.rotated_coder import RotatedCoder
    from nerf_rpn_amd.model.level_mapper import _setup_scales
    g = golden("detector")
    T = torch.from_numpy
    coder = RotatedCoder()
    assert torch.allclose(coder.encode_single(T(g["gt"]), T(g["rois"])), T(g["encoded"]), atol=1e-6)
    assert torch.allclose(coder.decode_single(T(g["deltas"]), T(g["rois"])), T(g["decoded"]), atol=1e-4, rtol=1e-6)
    mapper = _setup_scales([1 / 4, 1 / 8, 1 / 16, 1 / 32], 200, 4)
    assert torch.equal(mapper(T(g["mapper_boxes"])), T(g["mapper_levels"]))


def test_ngp_box_export_matches_reference(tmp_path):
    """scripts/proposals2ngp.py against boxes produced by the reference's functions (tests/golden/ngp_boxes.json) + the file round trip."""
    from nerf_rpn_amd.scripts import proposals2ngp as P
    g = json.load(open(os.path.join(GOLDEN, "ngp_boxes.json")))
    feats = {k: (np.array(v) if isinstance(v, list) else v) for k, v in g["features"].items()}
    for mitsuba in (0, 1):
        f = dict(feats, from_mitsuba=bool(mitsuba))
        for kind, fn, boxes in (("aabb", P.proposals_to_ngp_boxes, np.array(g["aabb"])), ("obb", P.obb_to_ngp_boxes, np.array(g["obb"]))):
            for mine, ref in zip(fn(boxes, f), g["boxes"][f"{kind}_{mitsuba}"]):
                for key in ("orientation", "position", "extents"):
                    assert np.allclose(mine[key], ref[key], atol=1e-12), (kind, mitsuba, key)
    os.makedirs(tmp_path / "scenes" / "s" / "train"); os.makedirs(tmp_path / "p"); os.makedirs(tmp_path / "f")
    json.dump({"frames": []}, open(tmp_path / "scenes" / "s" / "train" / "transforms.json", "w"))
    np.savez(tmp_path / "p" / "s.npz", proposal=np.array(g["obb"], dtype=np.float32), score=np.linspace(0.4, 0.9, 5).astype(np.float32))
    np.savez(tmp_path / "f" / "s.npz", from_mitsuba=False, **feats)
    P.main(["--bbox_format", "obb", "--dataset", "hypersim", "--dataset_path", str(tmp_path / "scenes"), "--features_path", str(tmp_path / "f"),
            "--proposals_path", str(tmp_path / "p"), "--output_dir", str(tmp_path / "o")])
    out = json.load(open(tmp_path / "o" / "s.json"))
    assert len(out["bounding_boxes"]) == 4 and out["bounding_boxes"][0]["score"] > out["bounding_boxes"][-1]["score"] > 0.5


def test_roipool_torch_paths_match_reference(golden):
    """The oracle's restatement of ROIPool(use_cuda=False) -- the reference CLI's default pooling (oracle/roipool.py) -- against outputs of the
    reference itself (tests/golden/make_golden.py::gen_roipool): bit-exact features for OBB 'pooling' / 'interpolation' and AABB crops, and the
    caller's OBB RoIs come back enlarged in place exactly as the reference leaves them (detector.py:195-201, 281).  The product path is
    csrc/roipool.hip (tests/test_gpu_detector.py holds it to the same vectors and to this oracle)."""
    from nerf_rpn_amd.model.detector import ROIPool
    from oracle.roipool import ROIPoolOracle
    g = golden("roipool")
    T = lambda k: torch.from_numpy(g[k])
    feats = [[T(f"feat{k}_{l}") for l in range(3)] for k in range(2)]
    scales = [int(v) for v in g["scales"]]
    for kind in ("pooling", "interpolation"):
        rois = T("obb_rois").clone()
        out = torch.stack(ROIPoolOracle([3, 3, 3], scales, 0.2, True, kind)(feats, rois))
        assert torch.equal(out, T("obb_" + kind)), kind
        assert torch.equal(rois, T("obb_rois_after_" + kind)) and not torch.equal(rois, T("obb_rois"))
    aabb = T("aabb_rois")
    out = torch.stack(ROIPoolOracle([2, 2, 2], scales, 0.2, False)(feats, [r for r in aabb]))
    assert torch.equal(out, T("aabb_pooling")) and torch.equal(aabb, T("aabb_rois"))
    with pytest.raises(NameError):
        ROIPool([2, 2, 2], scales, 0.2, True, "nearest", use_cuda=False)
    # the constructor default is the reference's use_cuda=False; the theta = 0 kernel path for AABBs is an explicit opt-in
    assert ROIPool([2, 2, 2], scales, 0.2, False).use_cuda is False and ROIPool([2, 2, 2], scales, 0.2, False).aabb_use_kernel is False

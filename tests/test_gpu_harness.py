"""GPU: metrics on the HIP IoU kernels vs golden values captured from the reference, and the command line end to end
(train one epoch on two tiny synthetic scenes, checkpoint, eval with proposal dump + eval.json)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_metrics_match_reference(golden, dev):
    from nerf_rpn_amd.eval import evaluate_box_proposals_ap, evaluate_box_proposals_recall
    g = golden("metrics")
    for tag in ("aabb", "obb"):
        P = [torch.from_numpy(g[f"{tag}_props{i}"]).to(dev) for i in range(3)]
        S = [torch.from_numpy(g[f"{tag}_scores{i}"]).to(dev) for i in range(3)]
        G = [torch.from_numpy(g[f"{tag}_gt{i}"]).to(dev) for i in range(3)]
        r50 = evaluate_box_proposals_recall(P, S, G, thresholds=torch.tensor([0.5]), limit=40)
        r25 = evaluate_box_proposals_recall(P, S, G, thresholds=torch.tensor([0.25]), limit=None)
        ar = evaluate_box_proposals_recall(P, S, G, thresholds=torch.arange(0.25, 1.0, 0.05), limit=100)
        assert abs(r50["ar"].item() - float(g[f"{tag}_r50"])) < 1e-6 and r50["num_pos"] == int(g[f"{tag}_num_pos"])
        assert abs(r25["ar"].item() - float(g[f"{tag}_r25"])) < 1e-6
        assert torch.allclose(ar["recalls"], torch.from_numpy(g[f"{tag}_ar_recalls"]), atol=1e-6)
        assert torch.allclose(ar["gt_overlaps"], torch.from_numpy(g[f"{tag}_gt_overlaps"]), atol=1e-5)
        assert abs(evaluate_box_proposals_ap(P, S, G, iou_thresh=0.5)["ap"].item() - float(g[f"{tag}_ap50"])) < 1e-6
        assert abs(evaluate_box_proposals_ap(P, S, G, iou_thresh=0.25, top_k=50)["ap"].item() - float(g[f"{tag}_ap25"])) < 1e-6


def test_device_metrics_on_proposal_sized_sets_match_the_oracle(dev):
    """recall matching (nrpn_recall_match_f32) and AP marking (nrpn_ap_mark) at the sizes the evaluation runs at -- 2500 proposals per
    scene, ground-truth boxes that overlap nothing (their column is all zeros: torch.max's first-index tie rule decides which proposal
    they retire), duplicated detections of one GT, a scene without proposals and one without ground truth -- against the CPU oracle
    (oracle/metrics.py, pinned to the reference by metrics.npz): recall counts exact, AP 1e-6."""
    from nerf_rpn_amd.eval import evaluate_box_proposals_ap, evaluate_box_proposals_recall
    from oracle import metrics as OM
    g = torch.Generator().manual_seed(17)

    def boxes(n, lo, hi, smin, smax, rot):
        c = torch.rand(n, 3, generator=g) * (hi - lo) + lo
        s = torch.rand(n, 3, generator=g) * (smax - smin) + smin
        if rot:
            return torch.cat([c, s, (torch.rand(n, 1, generator=g) - 0.5) * 3.1], dim=1)
        return torch.cat([c - s / 2, c + s / 2], dim=1)
    for rot in (True, False):
        P, S, G = [], [], []
        for sc, (n, ngt) in enumerate(((2500, 40), (1200, 7), (0, 5), (300, 0), (2500, 90))):
            gt = boxes(ngt, 20, 140, 8, 40, rot)
            parts = [boxes(max(n - 6 * min(ngt, 30), 0), 0, 160, 4, 30, rot)]
            if ngt and n:
                near = gt[:min(ngt, 30)].repeat_interleave(6, dim=0).clone()
                near[:, :3] += torch.randn(near.shape[0], 3, generator=g) * 1.5
                if not rot:
                    near[:, 3:] += torch.randn(near.shape[0], 3, generator=g) * 1.5
                    near[:, 3:] = torch.maximum(near[:, 3:], near[:, :3] + 1.0)
                parts.append(near)
            p = torch.cat(parts)[:n] if n else torch.zeros((0, 7 if rot else 6))
            if ngt > 3:
                gt[-3:, :3] += 500.0          # three boxes far outside: they overlap nothing
                if not rot:
                    gt[-3:, 3:] += 500.0
            P.append(p); S.append(torch.rand(p.shape[0], generator=g)); G.append(gt)
        for thr, limit in ((torch.tensor([0.5]), 300), (torch.arange(0.25, 1.0, 0.05), None)):
            got = evaluate_box_proposals_recall([p.to(dev) for p in P], [s.to(dev) for s in S], [b.to(dev) for b in G], thresholds=thr, limit=limit)
            ref = OM.recall(P, S, G, thr, limit)
            assert got["num_pos"] == ref["num_pos"]
            assert torch.equal((got["gt_overlaps"][:, None] >= thr[None]).sum(0), (ref["gt_overlaps"][:, None] >= thr[None]).sum(0))     # counts exact
            assert torch.allclose(got["gt_overlaps"], ref["gt_overlaps"], atol=1e-5) and abs(got["ar"].item() - ref["ar"].item()) < 1e-6
        keep = [i for i, b in enumerate(G) if b.shape[0]]        # the reference (and the oracle) index an empty IoU matrix for a scene without GT
        Pa, Sa, Ga = [P[i] for i in keep], [S[i] for i in keep], [G[i] for i in keep]
        for thr, top_k in ((0.25, None), (0.5, 300)):
            got = evaluate_box_proposals_ap([p.to(dev) for p in Pa], [s.to(dev) for s in Sa], [b.to(dev) for b in Ga], iou_thresh=thr, top_k=top_k)
            ref = OM.average_precision(Pa, Sa, Ga, thr, top_k)
            assert abs(got["ap"].item() - ref["ap"].item()) < 1e-6, (rot, thr, got["ap"].item(), ref["ap"].item())
            # here such a scene contributes false positives only
            more = evaluate_box_proposals_ap([p.to(dev) for p in P], [s.to(dev) for s in S], [b.to(dev) for b in G], iou_thresh=thr, top_k=top_k)
            assert more["ap"].item() <= got["ap"].item() + 1e-9


def test_command_line_train_eval_roundtrip(tmp_path, dev):
    from nerf_rpn_amd.run_rpn import main
    rng = np.random.default_rng(0)
    f, b = tmp_path / "features", tmp_path / "boxes"
    os.makedirs(f); os.makedirs(b)
    for s in ("a", "b"):
        np.savez(f / f"{s}.npz", rgbsigma=(rng.random((32, 32, 24, 4), dtype=np.float32) * 4 - 2))
        np.save(b / f"{s}.npy", np.array([[12., 12, 10, 8, 6, 6, 0.2], [20, 18, 12, 10, 8, 6, -0.5]], dtype=np.float32))
    split = tmp_path / "split.npz"
    np.savez(split, train_scenes=np.array(["a", "b"]), val_scenes=np.array(["a"]), test_scenes=np.array(["a", "b"]))
    common = ["--dataset_name", "front3d", "--features_path", str(f), "--boxes_path", str(b), "--dataset_split", str(split),
              "--backbone_type", "vgg_EF", "--rotated_bbox", "--normalize_density", "--save_path", str(tmp_path / "out"),
              "--rpn_pre_nms_top_n_test", "500", "--rpn_post_nms_top_n_test", "300"]
    main(["--mode", "train", "--num_epochs", "1", "--batch_size", "1", "--lr", "1e-4", "--log_interval", "1"] + common)
    ck = torch.load(tmp_path / "out" / "model_best.pt", map_location="cpu")
    assert set(ck) == {"epoch", "backbone_state_dict", "rpn_head_state_dict", "train_args"} and ck["epoch"] == 1
    assert len(ck["backbone_state_dict"]) == 135 and len(ck["rpn_head_state_dict"]) == 12
    main(["--mode", "eval", "--checkpoint", str(tmp_path / "out" / "model_best.pt"), "--output_proposals"] + common)
    z = np.load(tmp_path / "out" / "proposals" / "a.npz")
    assert set(z.files) == {"proposal", "score"} and z["proposal"].shape[1] == 7 and z["proposal"].dtype == np.float32
    js = json.load(open(tmp_path / "out" / "eval.json"))
    assert {"recall_50_top_300", "recall_25_top_300", "recall_ar_top_300", "ap_50", "ap_25"} <= set(js)


def test_fcos_command_line_train_eval_roundtrip(tmp_path, dev):
    """run_fcos.py drop-in: one epoch with the reference's train_fcos.sh flags (swin_t for speed), checkpoint keys, eval with
    proposal files in the FCOS key layout (proposals / scores / level_indices) and eval.json."""
    from nerf_rpn_amd.run_fcos import main
    rng = np.random.default_rng(1)
    f, b = tmp_path / "features", tmp_path / "boxes"
    os.makedirs(f); os.makedirs(b)
    for s in ("a", "b"):
        np.savez(f / f"{s}.npz", rgbsigma=(rng.random((40, 32, 24, 4), dtype=np.float32) * 4 - 2))
        np.save(b / f"{s}.npy", np.array([[14., 12, 10, 12, 8, 8, 0.2], [24, 18, 12, 14, 10, 8, -0.5]], dtype=np.float32))
    split = tmp_path / "split.npz"
    np.savez(split, train_scenes=np.array(["a", "b"]), val_scenes=np.array(["a"]), test_scenes=np.array(["a", "b"]))
    common = ["--dataset", "front3d", "--features_path", str(f), "--boxes_path", str(b), "--dataset_split", str(split),
              "--backbone_type", "swin_t", "--rotated_bbox", "--normalize_density", "--save_path", str(tmp_path / "out"),
              "--norm_reg_targets", "--centerness_on_reg", "--nms_thresh", "0.3", "--pre_nms_top_n", "400", "--fpn_post_nms_top_n", "300"]
    main(["--mode", "train", "--num_epochs", "1", "--batch_size", "2", "--lr", "3e-4", "--weight_decay", "1e-3", "--log_interval", "1",
          "--center_sampling_radius", "1.5", "--iou_loss_type", "iou"] + common)
    ck = torch.load(tmp_path / "out" / "model_best.pt", map_location="cpu")
    assert set(ck) == {"epoch", "backbone_state_dict", "fcos_state_dict", "train_args"} and ck["epoch"] == 1
    assert any(k.startswith("head.cls_tower.0.") for k in ck["fcos_state_dict"]) and "head.scales.4.scale" in ck["fcos_state_dict"]
    main(["--mode", "eval", "--checkpoint", str(tmp_path / "out" / "model_best.pt"), "--output_proposals", "--save_level_index",
          "--batch_size", "2"] + common)
    z = np.load(tmp_path / "out" / "proposals" / "a.npz")
    assert set(z.files) == {"proposals", "scores", "level_indices"} and z["proposals"].shape[1] == 7
    assert z["proposals"].shape[0] == z["scores"].shape[0] == z["level_indices"].shape[0] <= 300
    js = json.load(open(tmp_path / "out" / "eval.json"))
    assert {"recall_50_top_300", "recall_25_top_300", "recall_ar_top_300", "ap_50", "ap_25"} <= set(js)


def test_rpn_command_line_with_swin_backbone(tmp_path, dev):
    """run_rpn.py with --backbone_type swin_t (train.sh uses swin_s): one training epoch incl. stochastic depth + eval."""
    from nerf_rpn_amd.run_rpn import main
    rng = np.random.default_rng(2)
    f, b = tmp_path / "features", tmp_path / "boxes"
    os.makedirs(f); os.makedirs(b)
    for s in ("a", "b"):
        np.savez(f / f"{s}.npz", rgbsigma=(rng.random((40, 32, 24, 4), dtype=np.float32) * 4 - 2))
        np.save(b / f"{s}.npy", np.array([[14., 12, 10, 12, 8, 8, 0.2], [24, 18, 12, 14, 10, 8, -0.5]], dtype=np.float32))
    split = tmp_path / "split.npz"
    np.savez(split, train_scenes=np.array(["a", "b"]), val_scenes=np.array(["a"]), test_scenes=np.array(["a", "b"]))
    common = ["--dataset_name", "front3d", "--features_path", str(f), "--boxes_path", str(b), "--dataset_split", str(split),
              "--backbone_type", "swin_t", "--rotated_bbox", "--save_path", str(tmp_path / "out"),
              "--rpn_pre_nms_top_n_test", "500", "--rpn_post_nms_top_n_test", "300"]
    main(["--mode", "train", "--num_epochs", "2", "--batch_size", "1", "--log_interval", "1"] + common)
    ck = torch.load(tmp_path / "out" / "model_best.pt", map_location="cpu")
    assert "stages.2.1.attn.relative_position_index" in ck["backbone_state_dict"]
    main(["--mode", "eval", "--checkpoint", str(tmp_path / "out" / "model_best.pt"), "--output_proposals"] + common)
    assert np.load(tmp_path / "out" / "proposals" / "b.npz")["proposal"].shape[1] == 7


def test_device_ingest_matches_host_loader(tmp_path, dev):
    """ops.ingest_rgbsigma (on-disk (W,L,H,4) layout -> channels-last compute dtype with density_to_alpha on the GPU) against the
    host loader that restates the reference's load_single_scene (datasets.py:39-63); and the model consumes the
    channels-last-backed scene without a layout round trip, giving the same proposals."""
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.datasets import RawScene, _grid_from_npz
    rng = np.random.default_rng(3)
    g32 = (rng.random((24, 20, 16, 4), dtype=np.float32) * 8 - 4)
    g8 = rng.integers(0, 256, (12, 10, 8, 4), dtype=np.uint8)
    np.savez(tmp_path / "f.npz", rgbsigma=g32)
    np.savez(tmp_path / "u.npz", rgbsigma=g8)
    for path, norm in (("f.npz", True), ("f.npz", False), ("u.npz", False)):
        host = _grid_from_npz(str(tmp_path / path), norm)
        raw = _grid_from_npz(str(tmp_path / path), norm, raw=True)
        assert isinstance(raw, RawScene) and tuple(raw.shape) == tuple(host.shape)
        got = raw.to_device(torch.float32)
        assert got.shape == host.shape and got.permute(1, 2, 3, 0).is_contiguous()
        assert torch.allclose(got.cpu(), host, atol=2e-6, rtol=1e-6), (path, norm, (got.cpu() - host).abs().max())
    assert not isinstance(_grid_from_npz(str(tmp_path / "u.npz"), True, raw=True), RawScene)       # numpy-cast quirk: host path only
    relu = ops.ingest_rgbsigma(torch.from_numpy(g32).to(dev), 2, torch.float32).cpu()
    ref = torch.from_numpy(np.clip(1.0 - np.exp(-np.clip(g32[..., 3], 0, None) / 100.0), 0.0, 1.0))
    assert torch.allclose(relu[3], ref, atol=2e-6)
    # the model on a channels-last-backed scene == the model on the plain [4,W,L,H] tensor
    from test_gpu_e2e import build
    m = build(False, 160, dev).eval()
    raw = _grid_from_npz(str(tmp_path / "f.npz"), True, raw=True)
    with torch.no_grad():
        (_, p1, _), _, s1 = m([raw.to_device(torch.float32)])
        (_, p2, _), _, s2 = m([_grid_from_npz(str(tmp_path / "f.npz"), True).to(dev)])
    assert p1[0].shape == p2[0].shape and torch.allclose(s1[0], s2[0], atol=1e-5) and torch.allclose(p1[0], p2[0], atol=1e-3)


def test_voxel_score_heatmaps(tmp_path, dev):
    """``objectness_output_paths`` (run_rpn.py --output_voxel_scores; reference rpn.py:538-549): one .npz per scene with keys '0'..'3', the
    per-voxel maximum over the 13 anchors of each level's objectness logits, cropped to ceil(size / 2^(level+2))."""
    from test_gpu_e2e import build, scene
    m = build(True, 160, dev).eval()
    x = scene((48, 40, 32), 9).to(dev)
    path = str(tmp_path / "scores.npz")
    with torch.no_grad():
        m([x], objectness_output_paths=[path])
        # the same maps straight from the head, for comparison
        feats = m.backbone(x[None])
        logits, _ = m.rpn.head(list(feats))
    z = np.load(path)
    assert sorted(z.files) == ["0", "1", "2", "3"]
    for lvl in range(4):
        w, l, h = np.ceil(np.array([48, 40, 32]) / 2 ** (lvl + 2)).astype(int)
        assert z[str(lvl)].shape == (w, l, h)
        ref = logits[lvl][0].float().max(dim=0)[0][:w, :l, :h].cpu().numpy()
        assert np.allclose(z[str(lvl)], ref, atol=1e-5)


def test_bench_distributed_path_on_one_rank(tmp_path):
    """bench.py's multi-GPU code path (process group, exchange events, per-rank gathers, the gradient_exchange block of the JSON line)
    on a one-rank RCCL group (NRPN_FORCE_EXCHANGE=1): the scaling run of the driver must not be the first time that code executes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NRPN_FORCE_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    for mode in ("auto", "rs_ag"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                              "--no-probe", "--no-extras", "--exchange", mode], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.strip()]
        js = [l for l in lines if l.startswith("{")]
        # the driver's contract: rank 0 prints ONE JSON line on stdout.  RCCL writes its own banner there ("RCCL version : ...", "HIP version",
        # "Hostname", "Librccl path"); nothing of THIS package may (the trainer's start-up line about the chosen exchange goes to stderr)
        assert len(js) == 1 and not any("nerf_rpn_amd" in l for l in lines if not l.startswith("{")), lines[:6]
        line = json.loads(js[0])
        ex = line["gradient_exchange"]
        assert line["n_gpus"] == 1 and ex["buckets"] >= 2 and len(line["per_rank_ms_per_step"]) == 1
        # the comm-only arm: 3 modes x 3 bucket sizes timed on the real arena; 'auto' takes the fastest fp32 one, a forced mode is reported as forced
        assert len(ex["comm_only_ms"]) == 9 and all(v > 0 for v in ex["comm_only_ms"].values())
        if mode == "auto":
            # auto never picks the bf16 exchange: the winner is the fastest entry of the START-UP table (fp32 candidates only); the full table
            # of the bench is a second, independent measurement
            fp32 = ex["startup_comm_only_ms"]
            assert not any(k.startswith("a2a_bf16") for k in fp32) and len(fp32) == 6
            best = min(fp32, key=fp32.get)
            assert best.startswith(ex["mode"] + "@") and ex["mode_chosen_by"].startswith("comm-only")
        else:
            assert ex["mode"] == mode
        assert line["value"] > 20 and line["config"]["parallelism"].startswith("dp1")

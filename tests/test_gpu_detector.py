"""GPU: the second-stage network (model/detector.py) -- RoI <-> ground-truth assignment and sampling against what the reference
produced for the same inputs and numpy seed (tests/golden/detector.npz), ROIPool / RCNN / Classification_Model forward + backward, and
the run_rpn_detect.py command line round trip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def test_proposal_target_layer_matches_reference(golden, dev):
    from nerf_rpn_amd.model.detector import ProposalTargetLayer
    g = golden("detector")
    layer = ProposalTargetLayer(2, batch_size=64, fg_fraction=0.5, fg_threshold=0.35, bg_threshold=0.15, is_rotated_bbox=True)
    rois = [T(g["roi0"], dev), T(g["roi1"], dev)]
    gts = [T(g["gt0"], dev), T(g["gt1"], dev)]
    labels = [torch.ones(6, device=dev) for _ in range(2)]
    lab, r, gtr = layer(rois, gts, labels, is_sample=False)
    assert torch.equal(lab[0].cpu(), T(g["labels_all0"])) and torch.equal(lab[1].cpu(), T(g["labels_all1"]))
    assert torch.equal(gtr[0].cpu(), T(g["gt_rois_all0"]))
    np.random.seed(0)                                            # the sampling consumes numpy's global RNG like the reference
    lab, r, gtr = layer(rois, gts, labels, is_sample=True)
    assert torch.equal(lab.cpu(), T(g["labels_sampled"]))
    assert torch.equal(r.cpu(), T(g["rois_sampled"])) and torch.equal(gtr.cpu(), T(g["gt_rois_sampled"]))


def test_roipool_samples_the_box_it_is_given(dev):
    """A feature map that is 1 inside a rotated box and 0 outside: pooling THAT box (enlarge 0) gives ~1 in every bin, pooling the box
    rotated the other way gives clearly less -- the heading convention of the op (degrees, clockwise) is handled by ROIPool."""
    from nerf_rpn_amd.model.detector import ROIPool
    X = 48
    c, ext, th = torch.tensor([24., 22., 20.]), torch.tensor([30., 10., 12.]), 0.6
    ii = torch.stack(torch.meshgrid(*[torch.arange(X, dtype=torch.float32)] * 3, indexing="ij"), -1) - c
    lx = ii[..., 0] * np.cos(th) + ii[..., 1] * np.sin(th)
    ly = -ii[..., 0] * np.sin(th) + ii[..., 1] * np.cos(th)
    inside = ((lx.abs() <= ext[0] / 2) & (ly.abs() <= ext[1] / 2) & (ii[..., 2].abs() <= ext[2] / 2)).float()
    feat = inside[None].repeat(4, 1, 1, 1).to(dev)
    pool = ROIPool([3, 3, 3], [1], enlarge_scale=0.0, is_rotated_bbox=True, use_cuda=True)
    good = pool([[feat]], [torch.tensor([[0., 24., 22., 20., 26., 8., 10., th]], device=dev)])[0]
    bad = pool([[feat]], [torch.tensor([[0., 24., 22., 20., 26., 8., 10., -th]], device=dev)])[0]
    assert good.shape == (1, 4, 3, 3, 3) and good.min().item() > 0.95 and bad.mean().item() < 0.8


def test_roipool_default_paths_as_hip_kernels_match_the_reference(golden, dev):
    """Round 5 (f3): the reference's default pooling (use_cuda=False) runs on csrc/roipool.hip.  Against the reference's own CPU outputs
    (roipool.npz): AABB crops + max are integer / comparison work -- EXACT; the rotated paths evaluate sin / cos / divisions on the device, whose
    last bit can move a sample point across an integer: >= 99.9 % of the values within 1e-5, every value finite; the caller's OBB RoIs come
    back enlarged in place.  Dispatch as the reference (detector.py:239-245): axis-aligned RoIs take normal_forward whatever use_cuda says."""
    from nerf_rpn_amd.model.detector import ROIPool
    g = golden("roipool")
    feats = [[T(g[f"feat{k}_{l}"], dev) for l in range(3)] for k in range(2)]
    scales = [int(v) for v in g["scales"]]
    for kind in ("pooling", "interpolation"):
        rois = T(g["obb_rois"], dev).clone()
        out = torch.stack(ROIPool([3, 3, 3], scales, 0.2, True, kind, use_cuda=False)(feats, rois)).cpu()
        ref = T(g["obb_" + kind])
        close = ((out - ref).abs() <= 1e-5 + 1e-5 * ref.abs()).float().mean().item()
        assert out.shape == ref.shape and close >= 0.999 and torch.isfinite(out).all(), (kind, close)
        assert torch.allclose(rois.cpu(), T(g["obb_rois_after_" + kind]), rtol=1e-6)
    for kw in ({"use_cuda": False}, {}, {"use_cuda": True}):
        out = torch.stack(ROIPool([2, 2, 2], scales, 0.2, False, **kw)(feats, [r for r in T(g["aabb_rois"], dev)])).cpu()
        assert torch.equal(out, T(g["aabb_pooling"])), kw            # integer crops + max: exact


@pytest.mark.parametrize("kind,dtype", [("aabb", torch.float32), ("pooling", torch.float32), ("interpolation", torch.float32), ("aabb", torch.bfloat16),
                                        ("pooling", torch.bfloat16)])
def test_roipool_kernels_forward_and_backward_match_the_oracle(kind, dtype, dev):
    """csrc/roipool.hip against oracle/roipool.py (the torch restatement that is bit-exact against the reference, run here on the CPU) on
    random RoIs over a 3-level pyramid with 32 channels, RoIs reaching over the borders, crops clipped by the map, RoIs smaller than a voxel:
    pooled features and -- through autograd on both sides -- the gradient of every level.  AABB: exact features, gradients to 1e-6 (fixed-point
    scatter, 2^-44); rotated: 99.9 % within 1e-5 (a sample point within an ulp of an integer may take the other corner set).  bf16 maps: the
    oracle sees the same bf16-rounded values.  Two runs give bit-identical gradients."""
    from nerf_rpn_amd.model.detector import ROIPool
    from oracle.roipool import ROIPoolOracle
    gen = torch.Generator().manual_seed(5)
    C, scales, R = 32, [4, 8, 16], 40
    feats = [torch.randn(C, 72 // s, 64 // s, 48 // s, generator=gen) for s in scales]
    if dtype == torch.bfloat16:
        feats = [f.bfloat16().float() for f in feats]
    lv = torch.randint(0, 3, (R, 1), generator=gen).float()
    if kind == "aabb":
        lo = torch.rand(R, 3, generator=gen) * torch.tensor([50., 44., 30.]) + 1
        hi = lo + torch.rand(R, 3, generator=gen) * torch.tensor([40., 30., 24.]) + 0.5          # some reach past the map: the slice clips them
        rois = torch.cat([lv, lo, hi], dim=1)
        rot, out_size = False, [2, 3, 2]
    else:
        ctr = torch.rand(R, 3, generator=gen) * torch.tensor([72., 64., 48.])
        ext = torch.rand(R, 3, generator=gen) * 30 + 0.5
        ext[:4] = torch.rand(4, 3, generator=gen) * 2 + 0.3                                     # smaller than one level voxel
        th = (torch.rand(R, 1, generator=gen) - 0.5) * 3.0
        rois = torch.cat([lv, ctr, ext, th], dim=1)
        rot, out_size = True, [3, 2, 3]
    fe = kind if kind != "aabb" else "pooling"
    # oracle (CPU, autograd through torch ops)
    fo = [f.clone().requires_grad_(True) for f in feats]
    ro = rois.clone()
    oo = ROIPoolOracle(out_size, scales, 0.2, rot, fe)([fo], ro[None] if rot else [ro])[0]
    w = torch.randn(oo.shape, generator=gen)
    (oo * w).sum().backward()
    outs = []
    for rep in range(2):
        fh = [f.to(dev).to(dtype).requires_grad_(True) for f in feats]
        rh = rois.clone().to(dev)
        oh = ROIPool(out_size, scales, 0.2, rot, fe, use_cuda=False)([fh], rh[None] if rot else [rh])[0]
        assert oh.dtype == torch.float32 and tuple(oh.shape) == tuple(oo.shape)
        (oh * w.to(dev)).sum().backward()
        torch.cuda.synchronize()
        outs.append((oh.detach().cpu(), [f.grad.float().cpu() for f in fh]))
        if rot:
            assert torch.allclose(rh.cpu(), ro, rtol=1e-6)          # enlarged in place, like the oracle's / the reference's
    assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))       # deterministic
    oh, gh = outs[0]
    if kind == "aabb":
        assert torch.equal(oh, oo.detach())
    else:
        close = ((oh - oo.detach()).abs() <= 1e-5 + 1e-5 * oo.detach().abs()).float().mean().item()
        assert close >= 0.999, (kind, close)
    for l, (a, b) in enumerate(zip(gh, fo)):
        ref = b.grad
        tol = (1e-6 if dtype == torch.float32 else 1e-2) * max(1.0, ref.abs().max().item())
        frac = ((a - ref).abs() <= tol).float().mean().item()
        assert frac >= (1.0 if kind == "aabb" else 0.999), (kind, l, frac, (a - ref).abs().max().item())


def test_roipool_reference_op_quirks(dev):
    """reference_op_quirks=True reproduces the two behaviours of the reference's --use_cuda path that the default corrects: rows come out
    level by level (detector.py:250-259), and the heading is handed over in radians to an op that reads degrees.  With theta = 0 only the
    order differs: the quirk rows are the default rows gathered level-major."""
    from nerf_rpn_amd.model.detector import ROIPool
    g = torch.Generator().manual_seed(5)
    feats = [[torch.randn(16, s, s, s, generator=g).to(dev) for s in (16, 8, 4)]]
    box = torch.cat([torch.rand(30, 3, generator=g) * 40 + 12, torch.rand(30, 3, generator=g) * 20 + 4, torch.zeros(30, 1)], dim=1)
    rois = torch.cat([torch.randint(0, 3, (30, 1), generator=g).float(), box], dim=1).to(dev)
    a = ROIPool([3, 3, 3], [4, 8, 16], 0.2, True, use_cuda=True, reference_op_quirks=False)([feats[0]], [rois])[0]
    b = ROIPool([3, 3, 3], [4, 8, 16], 0.2, True, use_cuda=True, reference_op_quirks=True)([feats[0]], [rois])[0]
    order = torch.cat([torch.nonzero(rois[:, 0] == l).view(-1) for l in range(3)])
    assert torch.equal(b, a[order]) and not torch.equal(order, torch.arange(30, device=dev))
    rois[:, 7] = 0.5                                              # 0.5 rad read as 0.5 degrees: almost the unrotated sampling, not the box's
    c = ROIPool([3, 3, 3], [4, 8, 16], 0.2, True, use_cuda=True, reference_op_quirks=True)([feats[0]], [rois])[0]
    d = ROIPool([3, 3, 3], [4, 8, 16], 0.2, True, use_cuda=True, reference_op_quirks=False)([feats[0]], [rois])[0]
    assert (c - b).abs().mean().item() < 0.25 * (d[order] - b).abs().mean().item()


@pytest.mark.parametrize("use_cuda", [True, False])
def test_classification_model_forward_backward(dev, use_cuda):
    from nerf_rpn_amd.model.detector import Classification_Model, ProposalTargetLayer, RCNN, ROIPool
    from nerf_rpn_amd.model.feature_extractor import Bottleneck
    torch.manual_seed(0)
    np.random.seed(1)
    g = torch.Generator().manual_seed(3)
    feats = [[torch.randn(256, s, s, s, generator=g).to(dev) for s in (20, 10, 5, 3)]]
    gt = torch.tensor([[30., 28., 26., 20., 16., 14., 0.3], [52., 50., 40., 18., 22., 12., -0.7]], device=dev)
    rois = torch.cat([gt.repeat(30, 1) + torch.randn(60, 7, generator=g).to(dev) * torch.tensor([3, 3, 3, 2, 2, 2, 0.2], device=dev),
                      torch.rand(40, 7, generator=g).to(dev) * torch.tensor([70, 70, 70, 20, 20, 20, 1.], device=dev) + 4])
    rois = torch.cat([torch.randint(0, 4, (100, 1), generator=g).float().to(dev), rois], dim=1)
    model = Classification_Model(None, ProposalTargetLayer(2, batch_size=64, fg_threshold=0.35, bg_threshold=0.15, is_rotated_bbox=True),
                                 ROIPool([3, 3, 3], [4, 8, 16, 32], 0.2, is_rotated_bbox=True, use_cuda=use_cuda),
                                 RCNN(256, Bottleneck, 2, [3, 3, 3], is_add_layer=True, is_rotated_bbox=True, is_flatten=True),
                                 is_rotated_bbox=True).to(dev).train()
    (boxes, labels), probs, losses = model([rois], [gt], [torch.ones(2, device=dev)], feats)
    assert labels.shape == (1, 64) and probs[0].shape == (64, 2) and boxes[0].shape == (64, 7)
    loss = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]
    loss.backward()
    assert torch.isfinite(loss).item()
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    model.eval()
    with torch.no_grad():
        (props, labs), probs, _ = model([rois], [gt], [torch.ones(2, device=dev)], feats, is_sample=False, is_reg=True)
    assert props[0].shape == (100, 7) and labs[0].shape == (100,) and torch.isfinite(props[0]).all()
    assert int(labs[0].sum()) >= 20                              # the jittered copies of the ground truth are foreground


def test_detect_command_line_roundtrip(tmp_path, dev):
    """python run_rpn_detect.py --mode train / eval on a two-scene synthetic set with --fine_tune (raw grids -> backbone -> RoI head)."""
    from nerf_rpn_amd.run_rpn_detect import main
    rng = np.random.default_rng(0)
    for d in ("f", "b", "r"):
        (tmp_path / d).mkdir()
    names = ["a", "b"]
    for n in names:
        np.savez(tmp_path / "f" / f"{n}.npz", rgbsigma=rng.random((48, 40, 32, 4), dtype=np.float32), resolution=np.array([48, 40, 32]))
        gt = np.array([[20., 18., 14., 14., 12., 10., 0.2], [32., 24., 16., 10., 10., 12., -0.5]], dtype=np.float32)
        np.save(tmp_path / "b" / f"{n}.npy", gt)
        props = np.concatenate([np.repeat(gt, 20, 0) + rng.normal(0, 1, (40, 7)).astype(np.float32) * [2, 2, 2, 1, 1, 1, .1],
                                rng.random((30, 7), dtype=np.float32) * [40, 32, 24, 10, 10, 10, 1] + 3]).astype(np.float32)
        np.savez(tmp_path / "r" / f"{n}.npz", proposals=props, level_indices=rng.integers(0, 4, props.shape[0]).astype(np.float32))
    np.savez(tmp_path / "split.npz", train_scenes=np.array(names), val_scenes=np.array(names[:1]), test_scenes=np.array(names))
    common = ["--features_path", str(tmp_path / "f"), "--boxes_path", str(tmp_path / "b"), "--rois_path", str(tmp_path / "r"),
              "--dataset_split", str(tmp_path / "split.npz"), "--save_root", str(tmp_path / "out"), "--fine_tune", "--backbone_type", "vgg_EF",
              "--output_size", "3", "3", "3", "--spatial_scale", "4", "8", "16", "32", "--is_add_layer", "--is_flatten", "--rotated_bbox",
              "--cls_batch_size", "64", "--batch_size", "1", "--fg_threshold", "0.25", "--bg_threshold", "0.25", "--rotate_prob", "0",
              "--flip_prob", "0", "--rot_scale_prob", "0", "--filter_score_threhold", "0.0"]
    main(["--mode", "train", "--num_epochs", "1", "--log_interval", "1"] + common)
    out = tmp_path / "out" / "wandb_root" / "wandb_process"
    ck = torch.load(out / "epoch_0.pt", map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "backbone_state_dict", "RCNN_dict", "train_args", "optimizer_state_dict", "scheduler_state_dict"}
    assert "RCNN_cls_score.weight" in ck["RCNN_dict"] and "layer.0.weight" in ck["RCNN_dict"]
    main(["--mode", "eval", "--checkpoint", str(out / "epoch_0.pt"), "--output_proposals"] + common)
    assert (out / "eval.json").exists()
    z = np.load(out / "objectness" / "0" / "a.npz")
    assert z["proposal"].shape[1] == 7 and z["score"].shape[0] == z["proposal"].shape[0]

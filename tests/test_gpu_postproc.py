"""GPU parity: RPN post-processing / target-assignment kernels vs the golden vectors (reference outputs) and the oracle.
Integer outputs (sort order, keep indices, labels, anchors) must be bit-exact; IoU / boxes within the stated tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def test_sort_vertices_dropin(golden, dev):
    from nerf_rpn_amd import ops
    g = golden("geometry")
    out = ops.sort_vertices(T(g["sort_vertices"], dev), T(g["sort_mask"], dev), T(g["sort_num_valid"], dev))
    assert out.dtype == torch.int32 and tuple(out.shape) == g["sort_order"].shape
    assert torch.equal(out.cpu(), T(g["sort_order"]))


def test_rotated_iou_pairs(golden, dev):
    from nerf_rpn_amd import ops
    g = golden("geometry")
    b1, b2 = T(g["b1"], dev), T(g["b2"], dev)
    iou = ops.iou3d_pair(b1, b2).cpu()
    ref = T(g["iou3d"])
    err = (iou - ref).abs()
    assert err.max() < 1e-5, (err.max(), err.argmax(), b1[0, err.argmax()].tolist(), b2[0, err.argmax()].tolist())
    assert abs(iou[0, 0] - 1) < 1e-6 and abs(iou[0, 1] - 1 / 3) < 1e-6 and abs(iou[0, 2] - 0.1138) < 1e-4
    # symmetry property
    # symmetry on the random pairs (the reference itself is not symmetric on the hand-made degenerate pairs 0..8)
    assert (ops.iou3d_pair(b2, b1).cpu() - iou)[:, 9:].abs().max() < 5e-5
    m = ops.iou3d_matrix(b1[0, :40], b2[0, :50]).cpu()
    assert (m - T(g["obb_matrix"])).abs().max() < 1e-5
    a = ops.iou3d_matrix(T(g["aabb_a"], dev), T(g["aabb_b"], dev)).cpu()
    assert torch.allclose(a, T(g["aabb_iou"]), atol=1e-7)


def test_rotated_iou_analytic_cases_on_the_device(dev):
    """The closed-form pairs of tests/test_oracle_golden.py::analytic_iou_cases through the three HIP IoU paths (paired kernel, all-pairs
    kernel, the fused differentiable loss kernel's IoU output), both argument orders."""
    from nerf_rpn_amd import ops
    from test_oracle_golden import analytic_iou_cases
    b1, b2, want = analytic_iou_cases()
    a, b = b1.to(dev), b2.to(dev)
    assert torch.allclose(ops.iou3d_pair(a[None], b[None])[0].cpu(), want, atol=3e-6)
    assert torch.allclose(ops.iou3d_pair(b[None], a[None])[0].cpu(), want, atol=3e-6)
    assert torch.allclose(ops.iou3d_matrix(a, b).diagonal().cpu(), want, atol=3e-6)
    assert torch.allclose(ops.rotated_iou_loss(a, b, "giou")[1].cpu(), want, atol=3e-6)


def test_differentiable_iou_matches_oracle(golden, dev):
    from nerf_rpn_amd.model.rotated_iou import oriented_iou_loss as L
    from oracle import geometry as OG
    g = golden("geometry")
    b1, b2 = T(g["b1"])[:, 9:209], T(g["b2"])[:, 9:209]    # skip the hand-made degenerate pairs: the sort is discontinuous there
    a = b1.clone().to(dev).requires_grad_(True)
    l, _, _ = L.cal_giou_3d(a, b2.to(dev))
    l.sum().backward()
    c = b1.clone().requires_grad_(True)
    lo, _, _ = OG.giou_3d(c, b2)
    lo.sum().backward()
    assert torch.allclose(l.detach().cpu(), lo.detach(), atol=2e-5)
    gerr = (a.grad.cpu() - c.grad).abs()
    assert (gerr > 2e-3 + 1e-3 * c.grad.abs()).sum() <= 3, (gerr.max(), int((gerr > 2e-3 + 1e-3 * c.grad.abs()).sum()))
    d, _ = L.cal_diou_3d(b1.to(dev).requires_grad_(True), b2.to(dev))
    assert torch.allclose(d.detach().cpu(), T(g["diou_loss"])[:, 9:209], atol=2e-5)


def test_nms_keep_indices_exact(golden, dev):
    from nerf_rpn_amd.model import utils as U
    g = golden("nms")
    for tag in ("aabb", "obb"):
        b, s, l = T(g[tag + "_boxes"], dev), T(g[tag + "_scores"], dev), T(g[tag + "_levels"], dev)
        k = U.nms(b, s, float(g["thr"])).cpu()
        assert torch.equal(k, T(g[tag + "_keep"])), tag
        kb = U.batched_nms(b, s, l, float(g["thr"])).cpu()
        assert torch.equal(kb, T(g[tag + "_keep_batched"])), tag
        again = U.batched_nms(b[kb.to(dev)], s[kb.to(dev)], l[kb.to(dev)], float(g["thr"]))
        assert again.numel() == kb.numel()      # idempotence
    two = torch.tensor([[0, 0, 0, 4, 4, 4.], [1, 1, 1, 5, 5, 5.]], device=dev)
    assert torch.equal(U.nms(two, torch.tensor([0.2, 0.9], device=dev), 0.3).cpu(), T(g["two_keep"]))
    assert U.nms(torch.zeros(0, 6, device=dev), torch.zeros(0, device=dev), 0.3).numel() == 0
    one = U.nms(two[:1], torch.tensor([0.5], device=dev), 0.3)
    assert one.tolist() == [0]


def test_nms_large_vs_oracle(dev):
    from nerf_rpn_amd.model import utils as U
    from oracle import boxes as OB
    gen = torch.Generator().manual_seed(5)
    n = 2500
    c = torch.rand(n, 3, generator=gen) * 100 + 10
    sz = torch.rand(n, 3, generator=gen) * 20 + 4
    th = (torch.rand(n, 1, generator=gen) - 0.5) * 3
    obb = torch.cat([c, sz, th], dim=1)
    sc = torch.rand(n, generator=gen)
    lv = torch.randint(0, 4, (n,), generator=gen)
    ref = OB.nms_per_level(obb, sc, lv, 0.3)
    got = U.batched_nms(obb.to(dev), sc.to(dev), lv.to(dev), 0.3).cpu()
    assert torch.equal(got, ref)


def test_topk_order(dev):
    from nerf_rpn_amd import ops
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(300000, generator=gen)
    x[1000:1100] = 5.0       # ties at the top: index-ascending
    x[70000] = float("-inf")
    offs = [0, 250000, 299000, 300000]
    idx, val = ops.segmented_topk(x.to(dev), offs, 2500)
    idx, val = idx.cpu(), val.cpu()
    for s in range(3):
        seg = x[offs[s]:offs[s + 1]]
        k = min(2500, seg.numel())
        order = sorted(range(seg.numel()), key=lambda i: (-seg[i].item(), i))[:k]
        assert idx[s, :k].tolist() == [o + offs[s] for o in order], s
        assert torch.equal(val[s, :k], seg[order])
        assert (idx[s, k:] == -1).all()
    # massive ties: every element equal -> the first k indices
    y = torch.full((100000,), 0.25)
    idx, _ = ops.segmented_topk(y.to(dev), [0, 100000], 2500)
    assert idx[0].tolist() == list(range(2500))


def test_topk_long_segments_match_the_single_workgroup_kernel(dev):
    """Segments of >= 65536 scores take the multi-workgroup selection (histograms with integer atomics, collect, ties, sort): the same
    rows as the one-workgroup kernel behind the old entry point and as a stable descending sort, on quantised scores (thousands of ties
    around the k-th key), -inf runs, NaNs, and a segment one element over the switch."""
    import ctypes
    from nerf_rpn_amd import lib, ops
    gen = torch.Generator().manual_seed(11)
    n0 = 200 * 200 * 130 // 64 * 13 + 7                      # ~1.06 M: level 0 of the 200 x 200 x 130 benchmark scene
    x = torch.randn(n0 + 65537 + 300000 + 5000, generator=gen)
    x[:n0] = (x[:n0] * 64).round() / 64                        # ~500 distinct values
    x[n0 + 1000:n0 + 30000] = float("-inf")
    x[n0 + 65537 + 5:n0 + 65537 + 300000:7] = 0.5
    x[n0 + 65537 + 11] = float("nan")
    offs = [0, n0, n0 + 65537, n0 + 65537 + 300000, x.numel()]
    xd = x.to(dev)
    for k in (2500, 16384, 1000):
        idx, val = ops.segmented_topk(xd, offs, k)
        old_idx = torch.empty_like(idx)
        old_val = torch.empty_like(val)
        host = (ctypes.c_int64 * len(offs))(*offs)
        lib.call("segmented_topk_f32", xd.data_ptr(), ctypes.addressof(host), len(offs) - 1, k, old_idx.data_ptr(), old_val.data_ptr(),
                 torch.cuda.current_stream().cuda_stream)
        assert torch.equal(idx, old_idx) and torch.equal(val.nan_to_num(7.0), old_val.nan_to_num(7.0)), k
        for s_ in (0, 1, 3):                                   # segments without NaN: (score desc, index asc) = stable descending sort
            seg = x[offs[s_]:offs[s_ + 1]]
            kk = min(k, seg.numel())
            order = torch.sort(seg, descending=True, stable=True).indices[:kk]
            assert torch.equal(idx[s_, :kk].cpu().long(), order + offs[s_]), (k, s_)
    again, _ = ops.segmented_topk(xd, offs, 2500)
    first, _ = ops.segmented_topk(xd, offs, 2500)
    assert torch.equal(again, first)                           # atomics on integers only: run-to-run identical


def test_nms_prefilter_keeps_every_decision(dev):
    """The rotated mask kernel skips pairs whose IoU is provably 0 and compacts the rest: keep masks must equal a greedy pass over the
    full IoU matrix (same device IoU function, every pair) with the `not (iou <= thr)` rule -- on clustered boxes plus degenerate rows
    (zero / negative extents, NaN, inf, huge) that must bypass the prefilter, and for a negative threshold (prefilter off)."""
    from nerf_rpn_amd import ops
    gen = torch.Generator().manual_seed(3)
    n = 1500
    ctr = torch.rand(12, 3, generator=gen) * 120 + 20
    c = ctr[torch.randint(0, 12, (n,), generator=gen)] + torch.randn(n, 3, generator=gen) * 6
    sz = torch.rand(n, 3, generator=gen) * 24 + 3
    th = (torch.rand(n, 1, generator=gen) - 0.5) * 3.1
    b = torch.cat([c, sz, th], dim=1)
    b[5, 3] = 0.0
    b[17, 3:6] = 0.0
    b[18] = b[17]
    b[40, 4] = -3.0
    b[77, 0] = float("nan")
    b[90, 5] = float("inf")
    b[130, 1] = 3e18
    b[200] = b[199]
    lv = torch.sort(torch.randint(0, 3, (n,), generator=gen)).values.to(torch.int32)
    bd, ld = b.to(dev), lv.to(dev)
    iou = ops.iou3d_matrix(bd, bd).cpu()
    for thr in (0.3, 0.0, -0.5):
        keep = ops.nms3d_sorted(bd, ld, thr).cpu().bool()
        ref = torch.zeros(n, dtype=torch.bool)
        dead = torch.zeros(n, dtype=torch.bool)
        for i in range(n):
            if dead[i]:
                continue
            ref[i] = True
            sup = ~(iou[i] <= thr) & (lv == lv[i])
            sup[:i + 1] = False
            dead |= sup
        assert torch.equal(keep, ref), thr


def test_anchors_and_coders(golden, dev):
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.model.coder import AABBCoder, MidpointOffsetCoder
    g = golden("anchors")
    grids = [tuple(int(v) for v in x) for x in g["grids"]]
    tab = ops.AnchorTable(tuple(int(v) for v in g["mesh_size"]), grids, device=dev)
    assert torch.equal(ops.anchors(tab).cpu(), T(g["anchors"]))
    sel = torch.tensor([0, 5, 8190, tab.total - 1], device=dev)
    assert torch.equal(ops.anchors(tab, sel).cpu(), T(g["anchors"])[sel.cpu()])
    c = golden("coders")
    anc = T(c["anchors"], dev)
    ca, cm = AABBCoder(), MidpointOffsetCoder()
    assert torch.allclose(ca.encode_single(T(c["gt6"], dev), anc).cpu(), T(c["enc6"]), atol=2e-6)
    assert torch.allclose(ca.decode_single(T(c["d6"], dev), anc).cpu(), T(c["dec6"]), rtol=2e-6, atol=1e-4)
    assert torch.allclose(cm.encode_single(T(c["gt7"], dev), anc).cpu(), T(c["enc8"]), atol=1e-5)
    assert torch.allclose(cm.decode_single(T(c["d8"], dev), anc).cpu(), T(c["dec7"]), rtol=1e-5, atol=2e-4)
    assert torch.allclose(cm.decode_single_diff(T(c["d8"], dev), anc).cpu(), T(c["dec7"]), rtol=1e-5, atol=2e-4)
    assert torch.allclose(ops.obb_to_aabb(T(c["gt7"], dev)).cpu(), T(c["hbb"]), atol=2e-6)


def test_matcher_labels_exact(golden, dev):
    from nerf_rpn_amd import ops
    g = golden("matcher")
    a = golden("anchors")
    grids = [tuple(int(v) for v in x) for x in a["grids"]]
    tab = ops.AnchorTable(tuple(int(v) for v in a["mesh_size"]), grids, device=dev)
    gt = T(g["gt"], dev)
    labels, matched = ops.match_anchors(tab, ops.obb_to_aabb(gt), float(g["fg"]), float(g["bg"]))
    ref = T(g["matched"])
    ref_lab = (ref >= 0).float()
    ref_lab[ref == -2] = -1.0
    assert torch.equal(labels.cpu(), ref_lab)
    assert torch.equal(matched.cpu().long(), ref.clamp(min=0))
    # padded batch: anchors beyond ceil(ori/stride) are ignored
    ori = tuple(int(v) for v in a["ori_sizes"][1])
    lab2, _ = ops.match_anchors(tab, ops.obb_to_aabb(gt), float(g["fg"]), float(g["bg"]), ori)
    pm = T(a["padding_mask"])[1]
    assert (lab2.cpu()[~pm] == -1).all()


def test_sampled_loss_and_grad(dev):
    from nerf_rpn_amd import ops
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(2)
    M, dw = 5000, 8
    logits = torch.randn(M, generator=gen)
    deltas = torch.randn(M, dw, generator=gen)
    pos = torch.randperm(M, generator=gen)[:100].sort()[0]
    perm = torch.randperm(M, generator=gen)
    mask = torch.ones(M, dtype=torch.bool)
    mask[pos] = False
    neg = perm[mask[perm]][:156].sort()[0]          # disjoint from pos, as the sampler guarantees
    tgt = torch.randn(100, dw, generator=gen) * 0.3
    l1, d1 = logits.clone().requires_grad_(True), deltas.clone().requires_grad_(True)
    both = torch.cat([pos, neg])
    lab = torch.cat([torch.ones(100), torch.zeros(156)])
    ro = F.binary_cross_entropy_with_logits(l1[both], lab)
    rr = F.smooth_l1_loss(d1[pos], tgt, beta=1 / 9, reduction="sum") / both.numel()
    (ro + 5 * rr).backward()
    l2, d2 = logits.to(dev).requires_grad_(True), deltas.to(dev).requires_grad_(True)
    o, r = ops.SampledLossFn.apply(l2, d2, tgt.to(dev), pos.to(dev), neg.to(dev), 1 / 9)
    (o + 5 * r).backward()
    assert abs(o.item() - ro.item()) < 1e-6 and abs(r.item() - rr.item()) < 1e-6
    assert (l2.grad.cpu() - l1.grad).abs().max() < 1e-7 and (d2.grad.cpu() - d1.grad).abs().max() < 1e-6

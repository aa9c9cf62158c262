"""Sampled-anchor cones of the RPN head (csrc/cone.hip, the row-list conv kernels, ops.ConeHeadFn).

The training loss reads the head at the sampled anchors only (reference model/rpn.py:389-420), so the HIP path evaluates the head on the
receptive-field cones of those voxels.  Checked here: the voxel lists against the numpy oracle (exact), the row-list forward / dgrad /
wgrad kernels against plain torch fp32 convolutions on the CPU, and the cone head against the dense head of the same model (losses and
every parameter gradient), for one scene and for a padded batch of two."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fixture_init import seeded_state

pytestmark = pytest.mark.gpu

GRIDS = [(10, 8, 6), (5, 4, 3), (3, 2, 2)]
A = 13


def _sample(grids, n, rng, kp=40, kn=50):
    T = sum(g[0] * g[1] * g[2] for g in grids) * A
    pos = [np.sort(rng.choice(T, size=kp - 3 * i, replace=False)).astype(np.int64) for i in range(n)]
    neg = [np.sort(rng.choice(T, size=kn + 2 * i, replace=False)).astype(np.int64) for i in range(n)]
    return pos, neg, T


def _plan(grids, n, depth, pos, neg, dev):
    from nerf_rpn_amd import ops
    plan = ops.ConePlan(grids, n, depth, dev)
    mp, mn = max(len(p) for p in pos), max(len(q) for q in neg)
    out_pos = torch.full((n, mp + 3), 7, dtype=torch.int64, device=dev)       # slots beyond the counts hold junk the kernel must ignore
    out_neg = torch.full((n, mn + 5), 11, dtype=torch.int64, device=dev)
    cnt = []
    for i in range(n):
        out_pos[i, :len(pos[i])] = torch.from_numpy(pos[i]).to(dev)
        out_neg[i, :len(neg[i])] = torch.from_numpy(neg[i]).to(dev)
        cnt += [len(pos[i]), len(neg[i]), 0]
    counts = torch.tensor(cnt, dtype=torch.int32, device=dev)

    class Tab:      # what cone_build reads of ops.AnchorTable
        pass
    tab = Tab()
    tab.A = A
    tab.counts = [g[0] * g[1] * g[2] * A for g in grids]
    tab.offsets = [0] + list(np.cumsum(tab.counts))
    tab.total = tab.offsets[-1]
    plan.finish(ops.cone_build(plan, out_pos, out_neg, counts, tab).cpu().tolist())
    return plan


@pytest.mark.parametrize("grids,n,depth", [(GRIDS, 1, 3), (GRIDS, 2, 3), (GRIDS, 1, 0), ([(40, 40, 40), (20, 20, 20), (10, 10, 10), (5, 5, 5)], 1, 3),
                                           ([(12, 10, 8), (6, 5, 4), (3, 3, 2), (2, 2, 1)], 3, 2)])
def test_cone_lists_match_oracle(grids, n, depth, dev):
    from oracle import cone as OC
    rng = np.random.default_rng(3)
    big = grids[0][0] >= 40
    pos, neg, _ = _sample(grids, n, rng, 128 if big else 40, 128 if big else 50)
    plan = _plan(grids, n, depth, pos, neg, dev)
    want = OC.cone_lists(pos, neg, grids, A, depth)
    lists = plan.lists.cpu().numpy().view(np.uint32)
    prev = 0
    for k in range(depth + 1):
        ids, words = want[k]
        assert plan.counts[k] == len(ids), (k, plan.counts[k], len(ids))
        assert plan.counts[k] >= prev
        prev = plan.counts[k]
        np.testing.assert_array_equal(lists[k, :len(ids), 0], ids)
        np.testing.assert_array_equal(lists[k, :len(ids), 1], words)


def test_cone_build_flags_an_anchor_outside_the_pyramid(dev):
    rng = np.random.default_rng(4)
    pos, neg, T = _sample(GRIDS, 1, rng)
    pos[0][-1] = T + 5
    with pytest.raises(RuntimeError, match="outside the anchor pyramid"):
        _plan(GRIDS, 1, 2, pos, neg, dev)


def _level_views(t, grids, n):
    outs, off = [], 0
    for g in grids:
        c = g[0] * g[1] * g[2] * n
        outs.append(t[off:off + c].reshape(n, g[0], g[1], g[2], t.shape[-1]))
        off += c
    return outs


def _torch_conv(x_flat, w, b, grids, n, ksize):
    """plain torch fp32 on the CPU, level by level: [V, Cin] -> [V, Cout]"""
    outs = []
    for xl in _level_views(x_flat.float().cpu(), grids, n):
        y = F.conv3d(xl.permute(0, 4, 1, 2, 3), w.float().cpu(), None if b is None else b.float().cpu(), padding=ksize // 2)
        outs.append(y.permute(0, 2, 3, 4, 1).reshape(-1, w.shape[0]))
    return torch.cat(outs)


@pytest.mark.parametrize("dtype,ksize,cin,cout", [(torch.float32, 3, 64, 96), (torch.bfloat16, 3, 128, 256), (torch.bfloat16, 1, 256, 128),
                                                  (torch.float32, 1, 32, 128)])
def test_conv_rows_forward_matches_torch_on_the_listed_rows(dtype, ksize, cin, cout, dev):
    from nerf_rpn_amd import ops
    n = 2
    rng = np.random.default_rng(5)
    pos, neg, _ = _sample(GRIDS, n, rng)
    plan = _plan(GRIDS, n, 2, pos, neg, dev)
    V = plan.total
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(V, cin, generator=g) * 0.5).to(dtype)
    w = torch.randn(cout, cin, ksize, ksize, ksize, generator=g) / (cin * ksize ** 3) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    w_used = w.to(dtype).float()
    wp, wpd = ops.PackedWeight().get([w.to(dev)], dtype, cout, True, cin)
    mask = torch.randn(V, cout, generator=g).to(dtype)
    ref = _torch_conv(x, w_used, b, GRIDS, n, ksize)
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    for k in (0, 2):
        ids = plan.lists[k, :plan.counts[k], 0].long().cpu()
        for relu, use_mask, out_f32 in ((True, False, False), (False, True, False), (False, False, True)):
            odt = torch.float32 if out_f32 else dtype
            y = torch.full((V, cout), 3.0, dtype=odt, device=dev)
            ops.conv_rows_fwd(x.to(dev), wp, b.to(dev), y, plan, k, cin, cout, cout, ksize, lib_flags(relu), mask.to(dev) if use_mask else None)
            want = ref.clone()
            if relu:
                want = want.clamp_min(0)
            if use_mask:
                want = torch.where(mask.float() > 0, want, torch.zeros_like(want))
            got = y.float().cpu()
            err = (got[ids] - want[ids]).abs().max().item() / (want[ids].abs().max().item() + 1e-9)
            assert err < tol, (dtype, ksize, k, relu, use_mask, out_f32, err)
            rest = torch.ones(V, dtype=torch.bool)
            rest[ids] = False
            assert torch.all(got[rest] == 3.0)          # rows outside the list are not written


def test_conv_rows_long_list_runs_the_256x256_tile(dev):
    """The opt-in 256x256-tile form of the row-list conv (conv_igemm_big_kernel<ROWS>, tools switch nrpn_set_rows_big_tile; a measured
    negative, kept for A/B) on a list of >= 96 tiles of 256 rows, and the default 128-row tiling of the same list: listed rows against torch
    fp32 on the CPU, everything else untouched, with bias + ReLU and with a ReLU mask."""
    from nerf_rpn_amd import lib, ops
    grids = [(40, 40, 40)]
    rng = np.random.default_rng(8)
    pos, neg, _ = _sample(grids, 1, rng, 300, 300)
    plan = _plan(grids, 1, 3, pos, neg, dev)
    assert plan.counts[3] >= 96 * 256, plan.counts
    V, cin, cout = plan.total, 256, 256
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(V, cin, generator=g) * 0.5).clamp_min(0).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    wp, _ = ops.PackedWeight().get([w.to(dev)], torch.bfloat16, cout, True, cin)
    mask = torch.randn(V, cout, generator=g).to(torch.bfloat16)
    ref = _torch_conv(x, w.to(torch.bfloat16).float(), b, grids, 1, 3)
    ids = plan.lists[3, :plan.counts[3], 0].long().cpu()
    rest = torch.ones(V, dtype=torch.bool)
    rest[ids] = False
    try:
        for big in (1, 0):
            lib.call("set_rows_big_tile", big)
            for relu, use_mask in ((True, False), (False, True)):
                y = torch.full((V, cout), 3.0, dtype=torch.bfloat16, device=dev)
                ops.conv_rows_fwd(x.to(dev), wp, b.to(dev), y, plan, 3, cin, cout, cout, 3, lib_flags(relu), mask.to(dev) if use_mask else None)
                want = ref.clamp_min(0) if relu else torch.where(mask.float() > 0, ref, torch.zeros_like(ref))
                got = y.float().cpu()
                err = (got[ids] - want[ids]).abs().max().item() / want[ids].abs().max().item()
                assert err < 1.2e-2, (big, relu, use_mask, err)
                assert torch.all(got[rest] == 3.0)
    finally:
        lib.call("set_rows_big_tile", 0)


def lib_flags(relu):
    from nerf_rpn_amd import lib
    return lib.CONV_RELU if relu else 0


@pytest.mark.parametrize("dtype,ksize,cin,cout", [(torch.float32, 3, 64, 96), (torch.bfloat16, 3, 256, 256), (torch.bfloat16, 1, 256, 128),
                                                  (torch.float32, 3, 256, 256)])
def test_conv_rows_wgrad_matches_torch(dtype, ksize, cin, cout, dev):
    """dW, db from the listed rows only == autograd of the dense convolution when dY is zero outside the list; the 256 x 256 bf16 case runs
    conv_wgrad_big_kernel<ROWS>, the others conv_wgrad_kernel<ROWS>."""
    from nerf_rpn_amd import ops
    n = 2
    rng = np.random.default_rng(6)
    pos, neg, _ = _sample(GRIDS, n, rng)
    plan = _plan(GRIDS, n, 2, pos, neg, dev)
    V = plan.total
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(V, cin, generator=g) * 0.5).to(dtype)
    w = torch.randn(cout, cin, ksize, ksize, ksize, generator=g) / (cin * ksize ** 3) ** 0.5
    for k in (0, 1, 2):
        ids = plan.lists[k, :plan.counts[k], 0].long().cpu()
        dy = torch.zeros(V, cout)
        dy[ids] = torch.randn(len(ids), cout, generator=g)
        dy = dy.to(dtype)
        x_poison = x.clone()        # rows the list's taps never touch may hold anything (the cone buffers are torch.empty): NaN there
        if ksize == 1:
            rest = torch.ones(V, dtype=torch.bool)
            rest[ids] = False
            x_poison[rest] = float("nan")
        wt = w.clone().requires_grad_(True)
        bt = torch.zeros(cout, requires_grad=True)
        ref = _torch_conv(x, wt, bt, GRIDS, n, ksize)
        (ref * dy.float()).sum().backward()
        wparam = torch.nn.Parameter(w.to(dev))
        bparam = torch.nn.Parameter(torch.zeros(cout, device=dev))
        gw, gb = ops.conv_rows_wgrad(x_poison.to(dev), dy.to(dev), [wparam], [bparam], cout, ksize, plan, k)
        tol = 3e-5 if dtype == torch.float32 else 2e-3
        ew = (gw.cpu() - wt.grad).abs().max().item() / (wt.grad.abs().max().item() + 1e-9)
        eb = (gb.cpu() - bt.grad).abs().max().item() / (bt.grad.abs().max().item() + 1e-9)
        assert ew < tol and eb < tol, (dtype, ksize, k, ew, eb)


def rand_boxes(k, shape, rotated, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor([6.0, 6.0, 6.0])
    hi = torch.tensor([float(s) - 6.0 for s in shape])
    ctr = torch.rand(k, 3, generator=g) * (hi - lo) + lo
    size = torch.rand(k, 3, generator=g) * 14.0 + 5.0
    if rotated:
        return torch.cat([ctr, size, (torch.rand(k, 1, generator=g) - 0.5) * 3.0], dim=1)
    return torch.cat([(ctr - size / 2).clamp_min(0.0), torch.minimum(ctr + size / 2, torch.tensor([float(s) for s in shape]))], dim=1)


def _model(dev, rotated, dtype=torch.float32):
    from test_gpu_e2e import build
    m = build(rotated, 160, dev).train()
    m.set_compute_dtype(dtype)
    return m


@pytest.mark.parametrize("shapes,rotated", [([(48, 40, 32)], True), ([(48, 48, 32), (40, 32, 32)], False)])
def test_cone_head_matches_dense_head(shapes, rotated, dev):
    """Same model, same scenes, same injected sample: losses and every parameter gradient of the cone evaluation == the dense head's (fp32:
    the two differ only in the summation order of fp32 MFMA accumulations)."""
    from test_gpu_e2e import scene
    xs = [scene(s, 300 + i).to(dev) for i, s in enumerate(shapes)]
    gts = [rand_boxes(5, s, rotated, 90 + i).to(dev) for i, s in enumerate(shapes)]
    res = {}
    sample = {}
    for mode in ("dense", "cone"):
        m = _model(dev, rotated)
        m.rpn.use_cone = mode == "cone"
        if sample:
            m.rpn.sampler_hook = lambda labels: (sample["pos"], sample["neg"])
        torch.manual_seed(11)
        _, losses, _ = m(xs, gts)
        if not sample:
            sample["pos"], sample["neg"] = m.rpn.last_aux["pos"].clone(), m.rpn.last_aux["neg"].clone()
        (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]).backward()
        res[mode] = ({k: v.item() for k, v in losses.items()}, {k: p.grad.detach().float().cpu().clone() for k, p in m.named_parameters()})
    for k in ("loss_objectness", "loss_rpn_box_reg"):
        assert abs(res["cone"][0][k] - res["dense"][0][k]) < 2e-5 * max(1.0, abs(res["dense"][0][k])), (k, res["cone"][0][k], res["dense"][0][k])
    worst = 0.0
    top = max(g.abs().max().item() for g in res["dense"][1].values())
    for k, gd in res["dense"][1].items():
        gc = res["cone"][1][k]
        scale = gd.abs().max().item()
        if scale < 1e-6 * top:      # mathematically zero gradients (conv biases in front of a train-mode BatchNorm): rounding noise on both sides
            assert gc.abs().max().item() < 1e-5 * top, k
            continue
        err = (gc - gd).abs().max().item() / scale
        worst = max(worst, err)
        # train-mode BatchNorm over a few hundred voxels amplifies 1e-7 forward differences; the head / FPN tensors sit at 1e-5
        assert err < (2e-3 if "layers" in k or "bn" in k else 2e-4), (k, err)
    print(f"[cone vs dense] worst relative gradient difference {worst:.2e}")


def test_cone_head_bf16_trains_like_dense_bf16(dev):
    """bf16 (the bench dtype): cone and dense evaluate the same rows with different tile shapes, so single bf16 roundings differ; the losses
    must agree to bf16 resolution and the head's weight gradients in direction."""
    from test_gpu_e2e import scene
    shape = (64, 56, 48)
    x, gt = scene(shape, 400).to(dev), rand_boxes(6, shape, True, 91).to(dev)
    res, sample = {}, {}
    for mode in ("dense", "cone"):
        m = _model(dev, True, torch.bfloat16)
        m.rpn.use_cone = mode == "cone"
        if sample:
            m.rpn.sampler_hook = lambda labels: (sample["pos"], sample["neg"])
        torch.manual_seed(12)
        _, losses, _ = m([x], [gt])
        if not sample:
            sample["pos"], sample["neg"] = m.rpn.last_aux["pos"].clone(), m.rpn.last_aux["neg"].clone()
        (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]).backward()
        res[mode] = ({k: v.item() for k, v in losses.items()}, {k: p.grad.detach().float().cpu().clone() for k, p in m.rpn.head.named_parameters()})
    for k in ("loss_objectness", "loss_rpn_box_reg"):
        assert abs(res["cone"][0][k] - res["dense"][0][k]) < 2e-2 * max(1.0, abs(res["dense"][0][k])), (k, res["cone"][0][k], res["dense"][0][k])
    for k, gd in res["dense"][1].items():
        gc = res["cone"][1][k]
        cos = (gc.flatten() @ gd.flatten() / (gc.norm() * gd.norm() + 1e-30)).item()
        assert cos > 0.98, (k, cos)

"""GPU parity of the FCOS variant (csrc/fcos.hip + nerf_rpn_amd/model/fcos) against the oracle (oracle/fcos.py, pinned to the
reference's model/fcos/*.py) on identical inputs, and end to end against golden vectors captured from the reference."""
import argparse

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fixture_init import seeded_state

pytestmark = pytest.mark.gpu


def T(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-12)).item()


def fcos_args(rot, **kw):
    a = dict(num_convs=4, norm_reg_targets=True, centerness_on_reg=True, rotated_bbox=rot, pre_nms_thresh=0.0, pre_nms_top_n=2500,
             nms_thresh=0.3, fpn_post_nms_top_n=2500, min_size=0.0, center_sampling_radius=1.5, iou_loss_type="iou",
             use_additional_l1_loss=False, proj2d_loss_weight=0.0)
    a.update(kw)
    return argparse.Namespace(**a)


def build(rot, backbone, dev, **kw):
    from nerf_rpn_amd.model.feature_extractor import VGG_FPN, SwinTransformer_FPN
    from nerf_rpn_amd.model.fcos import FCOSOverNeRF
    if backbone == "swin":
        bb = SwinTransformer_FPN(patch_size=[4, 4, 4], embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=[4, 4, 4],
                                 stochastic_depth_prob=0, expand_dim=True)
    else:
        bb = VGG_FPN("EF", 4, True, 160)
    seeded_state(bb, 1)
    m = FCOSOverNeRF(fcos_args(rot, **kw), bb, [4, 8, 16, 32])
    seeded_state(m.fcos_module.head, 2, bias_jitter=0.5)
    for l, sc in enumerate(m.fcos_module.head.scales):
        sc.scale.data.fill_(0.8 + 0.15 * l)
    return m.to(dev)


def scene(shape, seed):
    return torch.rand(4, *[int(s) for s in shape], generator=torch.Generator().manual_seed(seed))


def test_groupnorm_matches_torch(dev):
    from nerf_rpn_amd import ops
    g = torch.Generator().manual_seed(0)
    for shape, groups, relu in [((2, 6, 5, 4, 256), 32, True), ((1, 9, 7, 3, 256), 32, False), ((3, 4, 4, 4, 64), 8, True)]:
        x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(dev).requires_grad_()
        w = (torch.rand(shape[-1], generator=g) + 0.5).to(dev).requires_grad_()
        b = (torch.randn(shape[-1], generator=g) * 0.3).to(dev).requires_grad_()
        dy = torch.randn(shape, generator=g).to(dev)
        y = ops.GroupNormFn.apply(x, w, b, groups, 1e-5, relu)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
        yr = F.group_norm(x.permute(0, 4, 1, 2, 3), groups, w, b, 1e-5)
        yr = (F.relu(yr) if relu else yr).permute(0, 2, 3, 4, 1)
        rx, rw, rb = torch.autograd.grad(yr, (x, w, b), dy)
        assert rel(y, yr) < 5e-6 and rel(gx, rx) < 2e-5 and rel(gw, rw) < 2e-5 and rel(gb, rb) < 2e-5, (shape, rel(y, yr), rel(gx, rx))


def test_groupnorm_fast_elementwise_passes_are_bit_identical(dev):
    """Round 6: the hoisted-parameter GroupNorm apply / backward-apply kernels (bf16, 8 channels per lane, C / groups % 8 == 0) against the
    general kernels: outputs and input gradients must be the same bits (two samples, ragged row counts, with and without ReLU); and the bf16
    result stays within bf16 rounding of torch's fp32 GroupNorm."""
    from nerf_rpn_amd import lib, ops
    g = torch.Generator().manual_seed(1)
    for shape, groups, relu in [((2, 6, 5, 4, 256), 32, True), ((1, 21, 7, 3, 256), 32, False), ((2, 9, 4, 4, 128), 16, True)]:
        x = (torch.randn(shape, generator=g) * 1.5 + 0.3).to(dev).bfloat16()
        w = (torch.rand(shape[-1], generator=g) + 0.5).to(dev)
        b = (torch.randn(shape[-1], generator=g) * 0.3).to(dev)
        dy = torch.randn(shape, generator=g).to(dev).bfloat16()
        out = []
        for fast in (1, 0):
            lib.call("set_gn_fast", fast)
            try:
                xx = x.clone().requires_grad_()
                y = ops.GroupNormFn.apply(xx, w, b, groups, 1e-5, relu)
                (gx,) = torch.autograd.grad(y, (xx,), dy)
                out.append((y.detach().clone(), gx.clone()))
            finally:
                lib.call("set_gn_fast", 1)
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]), shape
        yr = F.group_norm(x.float().permute(0, 4, 1, 2, 3), groups, w, b, 1e-5)
        yr = (F.relu(yr) if relu else yr).permute(0, 2, 3, 4, 1)
        assert rel(out[0][0].float(), yr) < 1e-2


@pytest.mark.parametrize("rot,ctr_on_reg,training", [(False, True, True), (True, True, False), (True, False, True)])
def test_head_matches_oracle_on_identical_features(rot, ctr_on_reg, training, dev):
    """FCOSHead forward + backward (towers, GroupNorm, fused final GEMMs, Scale / ReLU / stride epilogue) vs the oracle."""
    from nerf_rpn_amd.model.fcos import FCOSHead
    from oracle import fcos as OF
    hd = FCOSHead(256, 4, [4, 8, 16, 32], True, ctr_on_reg, rot)
    seeded_state(hd, 2, bias_jitter=0.5)
    for l, sc in enumerate(hd.scales):
        sc.scale.data.fill_(0.8 + 0.15 * l)
    orc = OF.FCOSHead(256, 4, [4, 8, 16, 32], True, ctr_on_reg, rot)
    orc.load_state_dict(hd.state_dict())
    hd = hd.to(dev).train(training)
    orc.train(training)
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(2, 256, *s, generator=g) for s in ((10, 8, 6), (6, 5, 4), (4, 4, 3), (3, 3, 3))]
    fo = [f.clone().requires_grad_() for f in feats]
    fg = [f.to(dev).requires_grad_() for f in feats]
    oo, og = orc(fo), hd(fg)
    flat_o = [t for grp in oo for t in grp]
    flat_g = [t for grp in og for t in grp]
    for a, r in zip(flat_g, flat_o):
        assert a.shape == r.shape and rel(a, r) < 2e-5, (a.shape, rel(a, r))
    dys = [torch.randn(t.shape, generator=g) * (torch.rand(t.shape, generator=g) < 0.2) for t in flat_o]
    used_o = [p for n, p in orc.named_parameters() if "scales.4" not in n]
    used_g = [p for n, p in hd.named_parameters() if "scales.4" not in n]
    go = torch.autograd.grad(flat_o, fo + used_o, dys)
    gg = torch.autograd.grad(flat_g, fg + used_g, [d.to(dev) for d in dys])
    names = [f"feat{i}" for i in range(4)] + [n for n, _ in orc.named_parameters() if "scales.4" not in n]
    for n, a, r in zip(names, gg, go):
        # four stacked GroupNorms over groups of a few hundred elements amplify fp32 summation-order differences (the split-K
        # partial sums of the small levels are added with atomics, so even two runs of this binary differ in the last bits)
        assert rel(a, r) < 5e-3, (n, rel(a, r))


@pytest.mark.parametrize("name", ["fcos_train_aabb_vgg", "fcos_train_aabb_giou_batch2", "fcos_train_obb_l1_proj"])
def test_targets_match_reference(name, golden, dev):
    """labels (exact) and regression targets of every location against the reference's prepare_targets."""
    from nerf_rpn_amd import ops
    g = golden(name)
    rot = bool(g["rotated"])
    shapes = [tuple(int(v) for v in s) for s in g["shapes"]]
    big = [max(s[d] for s in shapes) for d in range(3)]
    m = build(rot, "vgg", dev)
    with torch.no_grad():
        feats = m.backbone(torch.zeros(1, 4, *big, device=dev))
    geom = ops.FcosGeometry(len(shapes), [f.shape[-3:] for f in feats], [4, 8, 16, 32])
    gts = [T(g[f"gt{i}"], dev) for i in range(len(shapes))]
    labels, reg_t, npos = ops.fcos_targets(geom, gts, shapes if len(shapes) > 1 else None, 1.5, True, 8 if rot else 6, dev)
    keep = labels >= 0
    assert torch.equal(labels[keep].cpu(), T(g["labels"]))
    assert int(npos.item()) == int((T(g["labels"]) > 0).sum())
    ref = T(g["reg_targets"])
    pos = T(g["pos"])
    got = reg_t[keep].cpu()
    assert torch.allclose(got[pos], ref[pos], atol=2e-5, rtol=1e-5), (got[pos] - ref[pos]).abs().max()


@pytest.mark.parametrize("name", ["fcos_eval_aabb_vgg", "fcos_eval_obb_swin", "fcos_eval_obb_batch2", "fcos_eval_aabb_thresh"])
def test_eval_matches_reference(name, golden, dev):
    g = golden(name)
    rot = bool(g["rotated"])
    extra = {k: float(g[k]) for k in ("pre_nms_thresh", "min_size") if k in g}
    m = build(rot, str(g["backbone"]), dev, pre_nms_top_n=int(g["pre_nms_top_n"]), fpn_post_nms_top_n=int(g["fpn_post_nms_top_n"]), **extra).eval()
    xs = [scene(s, 300 + i).to(dev) for i, s in enumerate(g["shapes"])]
    with torch.no_grad():
        boxes, losses, scores = m(xs)
    assert losses == {}
    for i in range(len(xs)):
        rp, rs = T(g[f"boxes{i}"]), T(g[f"scores{i}"])
        gp, gs = boxes[i].cpu(), scores[i].cpu()
        # Random-weight heads emit many zero distances after the ReLU, i.e. OBBs of width ~1e-6 whose IoU is 0/0-like: the
        # reference's own NMS decisions on those flip between CPUs (the oracle run on the GPU box's host keeps 293 of scene 1
        # in fcos_eval_obb_batch2, the build container 296).  Hence a small allowance on the count and on unmatched rows.
        allow = max(5, rp.shape[0] // 50)
        assert abs(gp.shape[0] - rp.shape[0]) <= allow, (name, gp.shape, rp.shape)
        # score-descending lists; near-tied scores may swap and an IoU within 1e-6 of the NMS threshold may flip a decision
        # (cf. test_gpu_e2e.py): match every reference row to a row of equal level / score / box.
        near = (gs[None, :] - rs[:, None]).abs() <= 3e-6
        diff = (gp[None, :, 1:] - rp[:, None, 1:]).abs()
        tol = 3e-3 + 2e-4 * rp[:, 1:].abs()[:, None, :]
        ok = ((diff <= tol).all(dim=2) & near & (gp[None, :, 0] == rp[:, None, 0])).any(dim=1)
        assert (~ok).sum() <= allow, (name, i, int((~ok).sum()), rp.shape[0])


@pytest.mark.parametrize("name", ["fcos_train_aabb_vgg", "fcos_train_aabb_giou_batch2", "fcos_train_obb_swin", "fcos_train_obb_l1_proj",
                                  "fcos_train_obb_diou", "fcos_train_obb_smoothl1"])
def test_train_matches_reference(name, golden, dev):
    g = golden(name)
    rot = bool(g["rotated"])
    m = build(rot, str(g["backbone"]), dev, iou_loss_type=str(g["iou_loss_type"]), use_additional_l1_loss=bool(g["use_additional_l1_loss"]),
              proj2d_loss_weight=float(g["proj2d_loss_weight"])).train()
    xs = [scene(s, 400 + i).to(dev) for i, s in enumerate(g["shapes"])]
    gts = [T(g[f"gt{i}"], dev) for i in range(len(xs))]
    _, losses, _ = m(xs, gts)
    aux = m.fcos_module.loss_evaluator.last_aux
    assert torch.equal(aux["labels"][aux["labels"] >= 0].cpu(), T(g["labels"]))
    for k in ("loss_cls", "loss_reg", "loss_centerness"):
        ref = float(g[k])
        assert abs(losses[k].item() - ref) < 2e-4 * max(1.0, abs(ref)), (name, k, losses[k].item(), ref)
    (losses["loss_cls"] + losses["loss_reg"] + losses["loss_centerness"]).backward()
    params = dict(m.backbone.named_parameters())
    params.update({"head." + k: v for k, v in m.fcos_module.head.named_parameters()})
    unused = set(str(g["unused"]).split(","))
    flat_ref, flat_got = [], []
    # same acceptance rule as tests/test_gpu_e2e.py::test_train_matches_reference: each tensor within max(10 % of its scale,
    # 4 x the reference's own fp32 rounding error measured against float64) and the global direction within cos > 0.995
    for k, p in params.items():
        if k in unused:
            continue
        assert p.grad is not None, k
        if "grad/" + k in g:
            ref, got = T(g["grad/" + k]), p.grad.cpu()
        else:
            ref, got = T(g["gval/" + k]), p.grad.reshape(-1)[T(g["gidx/" + k], dev)].cpu()
        scale = float(g["gmax64/" + k])
        err = (got - ref).abs().max().item()
        allowed = max(0.1 * scale, 4.0 * float(g["err32/" + k])) + 5e-5
        assert err <= allowed, (name, k, err, allowed, scale)
        if scale > 1e-6:
            flat_ref.append(ref.reshape(-1) / scale)
            flat_got.append(got.reshape(-1) / scale)
    a, b = torch.cat(flat_ref).double(), torch.cat(flat_got).double()
    assert (a @ b / (a.norm() * b.norm())).item() > 0.995, name


def test_empty_targets_and_bf16(dev):
    """a scene without GT (all-background focal loss, zero regression terms) and the bf16 compute path run and stay finite."""
    m = build(True, "vgg", dev).train()
    x = scene((48, 40, 32), 1).to(dev)
    _, losses, _ = m([x], [torch.zeros(0, 7, device=dev)])
    assert losses["loss_reg"].item() == 0 and losses["loss_centerness"].item() == 0 and losses["loss_cls"].item() > 0
    sum(losses.values()).backward()
    m.set_compute_dtype(torch.bfloat16)
    gt = torch.tensor([[24., 20., 16., 20., 12., 10., 0.3]], device=dev)
    _, l16, _ = m([x], [gt])
    m.set_compute_dtype(torch.float32)
    _, l32, _ = m([x], [gt])
    for k in l32:
        assert torch.isfinite(l16[k]) and abs(l16[k].item() - l32[k].item()) < 0.1 * max(1.0, abs(l32[k].item())), (k, l16[k], l32[k])


@pytest.mark.parametrize("backbone,dtype", [("vgg", torch.float32), ("swin", torch.bfloat16)])
def test_two_runs_give_bit_identical_gradients(backbone, dtype, dev):
    """No floating-point atomics on the FCOS / Swin training path either: the focal-loss sum and the Scale gradient are ordered workgroup
    partials, the padded-token bias gradient of window attention is per-unit partials summed in window order.  Two forward/backward
    runs on the same batch (a grid that needs window padding) agree bit for bit."""
    m = build(True, backbone, dev).train()
    m.set_compute_dtype(dtype)
    x = scene((40, 36, 44), 3).to(dev)
    gt = torch.tensor([[20., 18., 16., 14., 12., 10., 0.3], [12., 24., 30., 10., 9., 12., -0.8]], device=dev)
    grads, losses = [], []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        _, ls, _ = m([x], [gt])
        loss = ls["loss_cls"] + ls["loss_reg"] + ls["loss_centerness"]
        loss.backward()
        losses.append(torch.stack([ls[k].detach() for k in sorted(ls)]))
        grads.append(torch.cat([p.grad.reshape(-1).float() for p in m.parameters() if p.grad is not None]))
    assert torch.equal(losses[0], losses[1]), (losses[0] - losses[1]).abs().max().item()
    assert torch.equal(grads[0], grads[1]), (grads[0] - grads[1]).abs().max().item()


def test_bf16_training_keeps_the_gradient_of_the_fp32_step(golden, dev):
    """config 4 (Swin-S + FCOS, OBB) in bf16, the training pass of `fcos_train_obb_swin`: LayerNorm / GroupNorm networks do not amplify
    rounding (tests/golden/bf16_emulation.json: Swin-S features move by ~1 % under bf16 storage), the FCOS targets depend on locations only,
    so the bf16 step must reproduce the fp32 step's losses within 2 % (measured 0.6 %) and keep every GEMM weight's gradient within 5 % in
    norm (measured 0.975-1.018) and at cosine >= 0.97 (measured min 0.9905, median 0.9998)."""
    g = golden("fcos_train_obb_swin")
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        m = build(True, "swin", dev, iou_loss_type=str(g["iou_loss_type"]), use_additional_l1_loss=bool(g["use_additional_l1_loss"]),
                  proj2d_loss_weight=float(g["proj2d_loss_weight"])).train()
        m.set_compute_dtype(dt)
        xs = [scene(s, 400 + i).to(dev) for i, s in enumerate(g["shapes"])]
        gts = [T(g[f"gt{i}"], dev) for i in range(len(xs))]
        _, losses, _ = m(xs, gts)
        (losses["loss_cls"] + losses["loss_reg"] + losses["loss_centerness"]).backward()
        out[dt] = ({k: v.item() for k, v in losses.items()},
                   {k: p.grad.detach().float().reshape(-1).double() for k, p in m.named_parameters() if p.grad is not None and p.dim() > 1})
    (l32, g32), (l16, g16) = out[torch.float32], out[torch.bfloat16]
    cos = sorted(((g32[k] @ g16[k] / (g32[k].norm() * g16[k].norm() + 1e-30)).item(), k) for k in g32 if g32[k].norm() > 0)
    ratio = sorted((g16[k].norm() / g32[k].norm()).item() for k in g32 if g32[k].norm() > 0)
    print(f"[bf16 fcos train] losses fp32 {l32} bf16 {l16}; gradient cosine min {cos[0]} median {cos[len(cos) // 2][0]:.4f}; norm ratio {ratio[0]:.3f}..{ratio[-1]:.3f}")
    for k in ("loss_cls", "loss_reg", "loss_centerness"):
        assert abs(l16[k] - l32[k]) <= 0.02 * max(1.0, abs(l32[k])), (k, l16[k], l32[k])
    assert cos[0][0] >= 0.97, cos[0]
    assert 0.95 <= ratio[0] and ratio[-1] <= 1.05, (ratio[0], ratio[-1])

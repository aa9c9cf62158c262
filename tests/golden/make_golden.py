"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference under the
shims of _refshim.py) on seeded inputs, asserting the oracle agrees, and saving inputs + expected
outputs.  Runs only in the build container:  python tests/golden/make_golden.py

Fixtures are data (inputs, seeds, expected outputs); no reference source is stored.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refshim  # noqa: E402

_refshim.install()

from fixture_init import seeded_state  # noqa: E402
from oracle import anchors as OA, boxes as OB, coders as OC, geometry as OG, metrics as OM, nets as ON, rpn as OR  # noqa: E402

# reference modules
from model import anchor as R_anchor, utils as R_utils, rpn as R_rpn  # noqa: E402
from model.coder import AABBCoder, MidpointOffsetCoder  # noqa: E402
from model.coder import misc as R_misc  # noqa: E402
from model.feature_extractor import VGG_FPN, ResNet_FPN_256, Bottleneck, SwinTransformer_FPN  # noqa: E402
from model.nerf_rpn import NeRFRegionProposalNetwork  # noqa: E402
from model.rotated_iou import oriented_iou_loss as R_iou, box_intersection_2d as R_b2d  # noqa: E402
import run_rpn as R_run  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def close(a, b, tol, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= tol, f"oracle != reference for {what}: max err {err}"
    return err


def rand_obb(n, g, lo=4.0, hi=60.0, smin=2.0, smax=24.0):
    c = torch.rand(n, 3, generator=g) * (hi - lo) + lo
    s = torch.rand(n, 3, generator=g) * (smax - smin) + smin
    t = (torch.rand(n, 1, generator=g) - 0.5) * math.pi
    return torch.cat([c, s, t], dim=1)


def rand_aabb(n, g, lo=0.0, hi=60.0, smin=2.0, smax=24.0):
    c = torch.rand(n, 3, generator=g) * (hi - lo) + lo
    s = torch.rand(n, 3, generator=g) * (smax - smin) + smin
    return torch.cat([c - s / 2, c + s / 2], dim=1)


# --------------------------------------------------------------------------------------------
def gen_geometry():
    print("geometry")
    g = torch.Generator().manual_seed(11)
    K = 600
    b1 = rand_obb(K, g, 10, 30, 2, 20)
    b2 = rand_obb(K, g, 10, 30, 2, 20)
    b2[:150, :3] = b1[:150, :3] + (torch.rand(150, 3, generator=g) - 0.5) * 4      # heavy overlaps
    special1 = torch.tensor([[0, 0, 0, 3, 3, 3, 0.], [1, 1, 1, 2, 2, 2, 0.], [0, 0, 0, 3, 3, 3, 0.],
                             [5, 5, 5, 4, 2, 3, 0.3], [5, 5, 5, 4, 2, 3, 0.3], [0, 0, 0, 2, 2, 2, 0.],
                             [0, 0, 0, 4, 4, 4, 0.], [0, 0, 0, 4, 2, 2, 0.], [0, 0, 0, 2, 2, 2, 0.]])
    special2 = torch.tensor([[0, 0, 0, 3, 3, 3, 0.], [2, 1, 1, 2, 2, 2, 0.], [1, 1, 1, 2, 2, 2, math.pi / 3],
                             [5, 5, 5, 4, 2, 3, 0.3], [5, 5, 5, 2, 4, 3, 0.3 - math.pi / 2], [10, 10, 10, 2, 2, 2, 0.],
                             [0, 0, 0, 1, 1, 1, 0.7], [0, 0, 0, 2, 4, 2, math.pi / 2], [2, 0, 0, 2, 2, 2, 0.]])
    b1 = torch.cat([special1, b1]).unsqueeze(0)
    b2 = torch.cat([special2, b2]).unsqueeze(0)
    iou = R_iou.cal_iou_3d(b1, b2)
    gl, gi, _ = R_iou.cal_giou_3d(b1, b2)
    dl, _ = R_iou.cal_diou_3d(b1, b2)
    # the sort op's own I/O, captured inside the reference stack
    c1 = R_iou.box2corners_th(b1[..., [0, 1, 3, 4, 6]])
    c2 = R_iou.box2corners_th(b2[..., [0, 1, 3, 4, 6]])
    inters, mi = R_b2d.box_intersection_th(c1, c2)
    c12, c21 = R_b2d.box_in_box_th(c1, c2)
    verts, mask = R_b2d.build_vertices(c1, c2, c12, c21, inters, mi)
    nv = mask.int().sum(2).int()
    mean = (verts * mask.float().unsqueeze(-1)).sum(2, keepdim=True) / nv[..., None, None]
    vn = (verts - mean).float()
    order = R_b2d.sort_indices(verts, mask)
    e = [close(OG.iou_3d(b1, b2), iou, 1e-6, "iou3d"), close(OG.giou_3d(b1, b2)[0], gl, 1e-5, "giou"),
         close(OG.diou_3d(b1, b2)[0], dl, 1e-5, "diou")]
    print("   max err", e, " iou known answers:", iou[0, :3].tolist())
    assert abs(iou[0, 0] - 1) < 1e-6 and abs(iou[0, 1] - 1 / 3) < 1e-6 and abs(iou[0, 2] - 0.1138) < 1e-4
    a = rand_aabb(40, g)
    b = rand_aabb(300, g)
    m = R_utils.box_iou_3d(a, b)
    close(OB.aabb_iou_matrix(a, b), m, 0, "aabb iou")
    om = R_utils.box_iou_3d(b1[0, :40], b2[0, :50])
    close(OB.iou_matrix(b1[0, :40], b2[0, :50]), om, 1e-6, "obb matrix")
    save("geometry", b1=b1, b2=b2, iou3d=iou, giou_loss=gl, diou_loss=dl,
         sort_vertices=vn, sort_mask=mask, sort_num_valid=nv, sort_order=order.int(),
         aabb_a=a, aabb_b=b, aabb_iou=m, obb_matrix=om)


def ref_anchor_gen():
    return R_anchor.AnchorGenerator3D(R_run.anchor_sizes, R_run.aspect_ratios, is_normalized=False)


def gen_anchors():
    print("anchors")
    ag = ref_anchor_gen()
    mesh = torch.zeros(2, 4, 40, 36, 28)
    grids = [(10, 9, 7), (5, 5, 4), (3, 3, 2), (2, 2, 1)]
    feats = [torch.zeros(2, 8, *gr) for gr in grids]
    anchors, _ = ag(mesh, feats)
    ratio_order = []
    for r in R_run.aspect_ratios[0]:
        import itertools
        ratio_order += list(set(itertools.permutations(r)))
    assert tuple(ratio_order) == OA.RATIO_ORDER, ratio_order
    mine = torch.cat(OA.all_anchors((40, 36, 28), grids))
    close(mine, anchors[0], 0, "anchors")
    ori = [(40, 36, 28), (33, 20, 28)]
    masks = ag.get_padding_masks(mesh, feats, ori)
    flat = torch.cat([R_rpn.permute_and_flatten(m, *m.shape[:2], 1, *m.shape[2:]) for m in masks], dim=1).squeeze(-1)
    assert torch.equal(flat, OA.padding_masks((40, 36, 28), grids, ori))
    save("anchors", mesh_size=[40, 36, 28], grids=grids, anchors=anchors[0], ratio_order=ratio_order,
         ori_sizes=ori, padding_mask=flat)


def gen_coders():
    print("coders")
    g = torch.Generator().manual_seed(5)
    M = 500
    anc = rand_aabb(M, g, 0, 60, 4, 40)
    gt6 = rand_aabb(M, g, 0, 60, 4, 40)
    gt7 = rand_obb(M, g, 0, 60, 4, 40)
    d6 = torch.randn(M, 6, generator=g) * 0.5
    d6[:5, 3:] = 9.0  # exercise the log(2000) clamp
    d8 = torch.randn(M, 8, generator=g) * 0.6
    d8[:5, 3:6] = 6.0
    d8[5:10, 3:6] = -6.0
    ca, cm = AABBCoder(), MidpointOffsetCoder()
    e6, x6 = ca.encode_single(gt6, anc), ca.decode_single(d6, anc)
    e8, x7 = cm.encode_single(gt7, anc), cm.decode_single(d8, anc)
    hbb = R_misc.obb2hbb_3d(gt7)
    close(OC.aabb_encode(gt6, anc), e6, 1e-6, "aabb enc"); close(OC.aabb_decode(d6, anc), x6, 1e-3, "aabb dec")
    close(OC.midpoint_encode(gt7, anc), e8, 1e-5, "mid enc"); close(OC.midpoint_decode(d8, anc), x7, 2e-4, "mid dec")
    close(OC.obb3d_to_hbb(gt7), hbb, 1e-6, "hbb3d")
    save("coders", anchors=anc, gt6=gt6, gt7=gt7, d6=d6, d8=d8, enc6=e6, dec6=x6, enc8=e8, dec7=x7, hbb=hbb)


def gen_matcher():
    print("matcher")
    g = torch.Generator().manual_seed(9)
    grids = [(10, 9, 7), (5, 5, 4), (3, 3, 2), (2, 2, 1)]
    anc = torch.cat(OA.all_anchors((40, 36, 28), grids))
    gt = rand_obb(6, g, 6, 30, 6, 20)
    gt[0, 3:6] = torch.tensor([8., 8., 8.]); gt[0, 6] = 0.0; gt[0, :3] = torch.tensor([16., 16., 12.])  # exact anchor hit -> ties
    q = R_utils.batched_box_iou(R_misc.obb2hbb_3d(gt), anc, 4)
    m = R_utils.Matcher(0.35, 0.2, allow_low_quality_matches=True)
    idx = m(q.clone())
    close(OB.iou_matrix_chunked(OC.obb3d_to_hbb(gt), anc, 4), q, 0, "match quality")
    assert torch.equal(OB.match(q.clone(), 0.35, 0.2), idx)
    save("matcher", anchors=anc, gt=gt, quality=q, matched=idx, fg=0.35, bg=0.2)


def gen_nms():
    print("nms")
    g = torch.Generator().manual_seed(21)
    out = {}
    for tag, n, maker in (("aabb", 400, rand_aabb), ("obb", 160, rand_obb)):
        base = maker(n // 8, g, 8, 40, 4, 14)
        bx = base.repeat_interleave(8, dim=0)
        bx[:, :3] += torch.randn(n, 3, generator=g) * 1.5
        if tag == "aabb":
            bx[:, 3:] = bx[:, :3] + (base.repeat_interleave(8, 0)[:, 3:] - base.repeat_interleave(8, 0)[:, :3]) \
                * (1 + 0.2 * torch.rand(n, 3, generator=g))
        else:
            bx[:, 6] += torch.randn(n, generator=g) * 0.2
        sc = torch.rand(n, generator=g)
        lv = torch.randint(0, 4, (n,), generator=g)
        k1 = R_utils.nms(bx, sc, 0.3)
        k2 = R_utils.batched_nms(bx, sc, lv, 0.3)
        assert torch.equal(OB.greedy_nms(bx, sc, 0.3), k1) and torch.equal(OB.nms_per_level(bx, sc, lv, 0.3), k2)
        out.update({f"{tag}_boxes": bx, f"{tag}_scores": sc, f"{tag}_levels": lv, f"{tag}_keep": k1, f"{tag}_keep_batched": k2})
    two = torch.tensor([[0, 0, 0, 4, 4, 4.], [1, 1, 1, 5, 5, 5.]])
    out["two_keep"] = R_utils.nms(two, torch.tensor([0.2, 0.9]), 0.3)  # quirk B4 edge
    save("nms", thr=0.3, **out)


def gen_metrics():
    print("metrics")
    import eval as R_eval
    g = torch.Generator().manual_seed(31)
    out = {}
    for tag, maker in (("aabb", rand_aabb), ("obb", rand_obb)):
        props, scores, gts = [], [], []
        for sc in range(3):
            gt = maker(6 + sc, g, 8, 40, 5, 14)
            jit = gt.repeat_interleave(12, dim=0).clone()
            jit[:, :3] += torch.randn(jit.shape[0], 3, generator=g) * 2.0
            if tag == "aabb":
                jit[:, 3:] = jit[:, :3] + (gt.repeat_interleave(12, 0)[:, 3:] - gt.repeat_interleave(12, 0)[:, :3]) * (0.8 + 0.4 * torch.rand(jit.shape[0], 3, generator=g))
            else:
                jit[:, 6] += torch.randn(jit.shape[0], generator=g) * 0.15
            far = maker(30, g, 0, 60, 4, 12)
            p = torch.cat([jit, far])
            props.append(p); scores.append(torch.rand(p.shape[0], generator=g)); gts.append(gt)
        r50 = R_eval.evaluate_box_proposals_recall(props, scores, gts, thresholds=torch.tensor([0.5]), limit=40)
        r25 = R_eval.evaluate_box_proposals_recall(props, scores, gts, thresholds=torch.tensor([0.25]), limit=None)
        ar = R_eval.evaluate_box_proposals_recall(props, scores, gts, thresholds=torch.arange(0.25, 1.0, 0.05), limit=100)
        ap50 = R_eval.evaluate_box_proposals_ap(props, scores, gts, iou_thresh=0.5)
        ap25 = R_eval.evaluate_box_proposals_ap(props, scores, gts, iou_thresh=0.25, top_k=50)
        close(OM.recall(props, scores, gts, torch.tensor([0.5]), 40)["ar"], r50["ar"], 1e-6, "r50")
        close(OM.recall(props, scores, gts, torch.arange(0.25, 1.0, 0.05), 100)["recalls"], ar["recalls"], 1e-6, "ar")
        close(OM.average_precision(props, scores, gts, 0.5)["ap"], ap50["ap"], 1e-6, "ap50")
        close(OM.average_precision(props, scores, gts, 0.25, 50)["ap"], ap25["ap"], 1e-6, "ap25")
        for i in range(3):
            out[f"{tag}_props{i}"], out[f"{tag}_scores{i}"], out[f"{tag}_gt{i}"] = props[i], scores[i], gts[i]
        out.update({f"{tag}_r50": r50["ar"], f"{tag}_r25": r25["ar"], f"{tag}_ar": ar["ar"], f"{tag}_ar_recalls": ar["recalls"],
                    f"{tag}_gt_overlaps": ar["gt_overlaps"], f"{tag}_ap50": ap50["ap"], f"{tag}_ap25": ap25["ap"], f"{tag}_num_pos": r50["num_pos"]})
        print("   ", tag, "R50", float(r50["ar"]), "AR", float(ar["ar"]), "AP50", float(ap50["ap"]), "AP25", float(ap25["ap"]))
    save("metrics", **out)


def gen_cli():
    print("cli + datasets")
    import argparse, json, tempfile
    import datasets as R_ds

    class Captured(Exception):
        pass

    def grab(self, *a, **k):
        raise Captured(self)
    orig = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = grab
    try:
        R_run.parse_args()
    except Captured as c:
        parser = c.args[0]
    finally:
        argparse.ArgumentParser.parse_args = orig
    flags = []
    for a in parser._actions:
        if not a.option_strings or a.dest == "help":
            continue
        flags.append(dict(options=a.option_strings, dest=a.dest, default=a.default, choices=list(a.choices) if a.choices else None,
                          type=a.type.__name__ if a.type else None, action=type(a).__name__))
    json.dump(flags, open(os.path.join(HERE, "cli_flags.json"), "w"), indent=1)
    print("   ", len(flags), "flags")
    # datasets: deterministic synthetic scene files -> what the reference loader returns
    rng = np.random.default_rng(3)
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(d + "/f"); os.makedirs(d + "/b")
        g32 = (rng.random((12, 10, 8, 4), dtype=np.float32) * 10 - 5).astype(np.float32)
        g8 = (rng.random((9, 8, 7, 4)) * 255).astype(np.uint8)
        np.savez(d + "/f/s0.npz", rgbsigma=g32); np.savez(d + "/f/s1.npz", rgbsigma=g8)
        b = np.array([[3., 2, 1, 4, 3, 2, 0.3], [6, 5, 4, 2, 2, 3, -0.8]], dtype=np.float32)
        np.save(d + "/b/s0.npy", b); np.save(d + "/b/s1.npy", b[:1])
        ds = R_ds.Front3DRPNDataset(d + "/f", d + "/b", scene_list=["s0", "s1"], normalize_density=True)
        x0, b0, n0 = ds[0]
        x1, b1, n1 = ds[1]
        import random
        random.seed(5)
        xa, ba = R_ds.BaseDataset.augment_rpn_inputs(x0.clone(), b0.clone(), 1.0, 1.0, 1.0, True)
        sn = R_ds.ScanNetRPNDataset.density_to_alpha(np.linspace(-3, 400, 9))
    save("datasets", g32=g32, g8=g8, boxes=b, x0=x0, x1=x1, aug_x=xa, aug_boxes=ba, scannet_alpha=sn)


SWIN_S = dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24])      # run_rpn.py:283
# every --backbone_type of the reference CLI (run_rpn.py:274-292)
SWIN_VARIANTS = {"swin": SWIN_S, "swin_s": SWIN_S,
                 "swin_t": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24]),
                 "swin_b": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24]),
                 "swin_l": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48])}


def build_ref(rotated, resolution, reg_loss="smooth_l1", **kw):
    if str(kw.get("backbone")).startswith("swin"):
        bb = SwinTransformer_FPN(patch_size=[4, 4, 4], window_size=[4, 4, 4], stochastic_depth_prob=kw.get("sd", 0.1), expand_dim=True,
                                 **SWIN_VARIANTS[kw["backbone"]])
    else:
        bb = ResNet_FPN_256(Bottleneck, [3, 4, 6, 3], input_dim=4, is_max_pool=True) if kw.get("backbone") == "resnet" \
            else VGG_FPN("AF" if kw.get("backbone") == "vgg_AF" else "EF", 4, True, resolution)
    hd = R_anchor.RPNHead(256, 13, 4, rotate=rotated)
    seeded_state(bb, 1); seeded_state(hd, 2)
    return NeRFRegionProposalNetwork(bb, ref_anchor_gen(), hd, rpn_pre_nms_top_n_train=2500, rpn_pre_nms_top_n_test=kw.get("pre", 2500),
                                     rpn_post_nms_top_n_train=2500, rpn_post_nms_top_n_test=kw.get("post", 2500), rpn_nms_thresh=0.3,
                                     rpn_fg_iou_thresh=0.35, rpn_bg_iou_thresh=0.2, rpn_batch_size_per_mesh=256,
                                     rpn_positive_fraction=0.5, rpn_score_thresh=0.0, rotated_bbox=rotated, reg_loss_type=reg_loss)


def build_oracle(rotated, resolution, reg_loss="smooth_l1", **kw):
    if str(kw.get("backbone")).startswith("swin"):
        v = SWIN_VARIANTS[kw["backbone"]]
        bb = ON.SwinFPN(v["embed_dim"], v["depths"], v["num_heads"], kw.get("sd", 0.1))
    else:
        bb = ON.ResNetFPN() if kw.get("backbone") == "resnet" else ON.VGGFPN("AF" if kw.get("backbone") == "vgg_AF" else "EF", 4, resolution)
    hd = ON.RPNHead(256, 13, 4, rotated)
    seeded_state(bb, 1); seeded_state(hd, 2)
    return OR.Detector(bb, OR.RPN(hd, rotated=rotated, reg_loss_type=reg_loss, pre_nms_top_n=kw.get("pre", 2500),
                                  post_nms_top_n=kw.get("post", 2500)))


def scene(shape, seed):
    return torch.rand(4, *shape, generator=torch.Generator().manual_seed(seed))


def subsample(t, n=4096):
    f = t.reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n, f.numel()), dtype=torch.float64).long().clamp_(max=f.numel() - 1)
    return idx, f[idx]


def gen_eval():
    print("end-to-end eval")
    cases = [("eval_aabb_s2", False, 160, [(48, 48, 48)], {}),
             ("eval_obb_s2", True, 160, [(48, 40, 32)], {}),
             ("eval_obb_s1_cfg0", True, 64, [(16, 16, 16)], {"pre": 600}),       # BASELINE config[0] at 16^3
             # BASELINE config[0] at its stated size: one 64^3 grid with --resolution 64 (stride-1 stem, 64^3 / 32^3 / 16^3 / 8^3 maps,
             # 3 893 760 anchors, a 3.4 M-element level-0 top-k segment)
             ("eval_obb_64_cfg0", True, 64, [(64, 64, 64)], {}),
             ("eval_aabb_batch2", False, 160, [(48, 48, 32), (40, 32, 32)], {}),
             ("eval_resnet_obb", True, 160, [(64, 56, 48)], {"backbone": "resnet"}),
             # swin_s: token grids 20x14x12 -> 10x7x6 -> 5x4x3 -> 3x2x2: window padding on every stage, a stage whose
             # shift applies to one axis only (padded 8x4x4), and odd sizes in the patch merging
             ("eval_swin_obb", True, 160, [(80, 56, 48)], {"backbone": "swin"}),
             ("eval_swin_aabb_batch2", False, 160, [(64, 64, 48), (48, 40, 40)], {"backbone": "swin"}),
             # round 5: the remaining --backbone_type choices of the CLI (run_rpn.py:274-292), each compared once:
             # VGG11-"AF" (two extra pools: five stages), Swin-T (depth 6 third stage), Swin-B (embed 128: channel widths 128..1024, 64-byte
             # K-step rows), Swin-L (embed 192, heads 6/12/24/48)
             ("eval_vgg_af_obb", True, 160, [(48, 40, 32)], {"backbone": "vgg_AF"}),
             ("eval_swin_t_obb", True, 160, [(64, 56, 48)], {"backbone": "swin_t"}),
             # swin_b: the reference's own table (embed 128, heads 3/6/12/24) does not run -- its attention reshape raises; recorded as such
             ("eval_swin_b_obb", True, 160, [(64, 56, 48)], {"backbone": "swin_b", "expect_reference_error": True}),
             ("eval_swin_l_aabb", False, 160, [(64, 48, 48)], {"backbone": "swin_l"})]
    only = os.environ.get("GOLDEN_ONLY")
    for name, rot, res, shapes, kw in cases:
        if only and only not in name:
            continue
        ref = build_ref(rot, res, **kw).eval()
        xs = [scene(s, 100 + i) for i, s in enumerate(shapes)]
        if kw.get("expect_reference_error"):
            import json
            try:
                with torch.no_grad():
                    ref([x.clone() for x in xs])
                raise AssertionError(f"{name}: the reference was expected to fail")
            except RuntimeError as e:
                rec = {"backbone_type": kw["backbone"], "shapes": shapes, "reference_raises": type(e).__name__, "message": str(e).splitlines()[0]}
                json.dump(rec, open(os.path.join(HERE, name + "_reference_failure.json"), "w"), indent=1)
                print(f"   {name}: the reference raises {rec['reference_raises']}: {rec['message']}")
            continue
        orc = build_oracle(rot, res, **kw)
        orc.backbone.eval()
        with torch.no_grad():
            (feats, props, lvls), _, scores = ref([x.clone() for x in xs])
            (ofeats, oprops, olvls), _, oscores, aux = orc([x.clone() for x in xs])
        arrs = {"shapes": shapes, "rotated": rot, "resolution": res, "pre": kw.get("pre", 2500), "backbone": kw.get("backbone", "vgg")}
        for i, (f, of) in enumerate(zip(feats, ofeats)):
            close(of, f, 5e-4, f"{name} feat{i}")
            idx, val = subsample(f)
            arrs[f"feat{i}_shape"], arrs[f"feat{i}_idx"], arrs[f"feat{i}_val"] = list(f.shape), idx, val
        for i in range(len(xs)):
            assert props[i].shape == oprops[i].shape, (name, props[i].shape, oprops[i].shape)
            close(oprops[i], props[i], 2e-3, f"{name} proposals"); close(oscores[i], scores[i], 1e-5, f"{name} scores")
            arrs[f"proposals{i}"], arrs[f"scores{i}"], arrs[f"levels{i}"] = props[i], scores[i], lvls[i]
            print(f"   {name}[{i}]: {props[i].shape[0]} proposals, score range {scores[i].min():.4f}..{scores[i].max():.4f}")
        save(name, **arrs)


def record_grads(arrs, rp, op, p64):
    """Store the reference gradients (full if small, 256 samples otherwise), their fp64 counterparts and the reference's own
    fp32 rounding error; returns the worst oracle-vs-reference relative error."""
    worst = 0.0
    for k, p in rp.items():
        gr, go = p.grad, op[k].grad
        # conv biases feeding a train-mode BatchNorm have a mathematically-zero gradient (pure
        # rounding noise ~1e-6 in both implementations), hence the absolute floor.
        rel = max(0.0, (gr - go).abs().max().item() - 2e-5) / (gr.abs().max().item() + 1e-12)
        worst = max(worst, rel)
        arrs["gnorm/" + k] = gr.norm()
        g64 = p64[k].grad
        arrs["err32/" + k] = (gr.double() - g64).abs().max()
        arrs["gmax64/" + k] = g64.abs().max()
        if gr.numel() <= 512:
            arrs["grad/" + k] = gr
            arrs["grad64/" + k] = g64
        else:
            idx, val = subsample(gr, 256)
            arrs["gidx/" + k], arrs["gval/" + k] = idx, val
            arrs["gval64/" + k] = g64.reshape(-1)[idx]
    return worst


def gen_detector():
    """Second stage (reference detector.py, coder/rotated_coder.py, level_mapper.py, run_rpn_detect.py): the pieces that run on the CPU
    in the reference -- the rotated RoI coder, the FPN level mapper, RoI <-> ground-truth assignment / sampling, the CLI flag table."""
    print("detector")
    import argparse, json
    from model.coder.rotated_coder import RotatedCoder
    from model.level_mapper import _setup_scales
    from model import detector as R_det
    g = torch.Generator().manual_seed(21)
    rois, gt = rand_obb(200, g, 8, 150, 4, 40), rand_obb(200, g, 8, 150, 4, 40)
    gt[:100, :3] = rois[:100, :3] + (torch.rand(100, 3, generator=g) - 0.5) * 6
    coder = RotatedCoder()
    enc = coder.encode_single(gt, rois)
    deltas = torch.randn(200, 7, generator=g) * 0.3
    deltas[:5, 3:6] = 9.0                                        # exercises the log(2000) clamp
    dec = coder.decode_single(deltas, rois)
    mapper = _setup_scales([1 / 4, 1 / 8, 1 / 16, 1 / 32], 200, 4)
    boxes = rand_obb(300, g, 8, 150, 1, 180)
    levels = mapper(boxes)
    # RoI <-> GT assignment: rows = (level, box)
    layer = R_det.ProposalTargetLayer(2, batch_size=64, fg_fraction=0.5, fg_threshold=0.35, bg_threshold=0.15, is_rotated_bbox=True)
    scene_rois, scene_gt = [], []
    for k in range(2):
        gtb = rand_obb(6, g, 20, 120, 10, 40)
        r = rand_obb(150, g, 10, 140, 6, 50)
        r[:40] = gtb[torch.randint(0, 6, (40,), generator=g)] + torch.randn(40, 7, generator=g) * torch.tensor([2, 2, 2, 1.5, 1.5, 1.5, 0.1])
        r[:, 3:6] = r[:, 3:6].clamp_min(2.0)
        scene_rois.append(torch.cat([torch.randint(0, 4, (150, 1), generator=g).float(), r], dim=1))
        scene_gt.append(gtb)
    labels = [torch.ones(6) for _ in range(2)]
    orig_get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: "cpu"             # fifth shim: the layer does .to(tensor.get_device()), -1 on the CPU
    try:
        lab_all, rois_all, gtr_all = layer([r.clone() for r in scene_rois], scene_gt, labels, is_sample=False)
        np.random.seed(0)
        lab_s, rois_s, gtr_s = layer([r.clone() for r in scene_rois], scene_gt, labels, is_sample=True)
    finally:
        torch.Tensor.get_device = orig_get_device
    # flag table of the CLI
    import run_rpn_detect as R_cli

    class Captured(Exception):
        pass

    def grab(self, *a, **k):
        raise Captured(self)
    orig = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = grab
    try:
        R_cli.parse_args()
    except Captured as c:
        parser = c.args[0]
    finally:
        argparse.ArgumentParser.parse_args = orig
    flags = [dict(options=a.option_strings, dest=a.dest, default=a.default, choices=list(a.choices) if a.choices else None,
                  type=a.type.__name__ if a.type else None, action=type(a).__name__, nargs=a.nargs)
             for a in parser._actions if a.option_strings and a.dest != "help"]
    json.dump(flags, open(os.path.join(HERE, "cli_flags_detect.json"), "w"), indent=1)
    print("   ", len(flags), "flags")
    save("detector", rois=rois, gt=gt, encoded=enc, deltas=deltas, decoded=dec, mapper_boxes=boxes, mapper_levels=levels,
         roi0=scene_rois[0], roi1=scene_rois[1], gt0=scene_gt[0], gt1=scene_gt[1], labels_all0=lab_all[0], labels_all1=lab_all[1],
         gt_rois_all0=gtr_all[0], labels_sampled=lab_s, rois_sampled=rois_s, gt_rois_sampled=gtr_s)


def gen_roipool():
    """ROIPool without the op (reference detector.py:264-438, the CLI default ``use_cuda=False``): AABB integer crops + adaptive max-pool,
    OBB rotated 8-corner gather + max-pool / trilinear resize, and the in-place enlargement the OBB path leaves in the caller's RoIs."""
    print("roipool (torch paths)")
    from model import detector as R_det
    g = torch.Generator().manual_seed(33)
    C, scales = 6, [4, 8, 16]
    feats = [[torch.randn(C, 80 // s, 64 // s, 48 // s, generator=g) for s in scales] for _ in range(2)]
    orig_get_device = torch.Tensor.get_device
    torch.Tensor.get_device = lambda self: "cpu"
    out = {}
    try:
        # OBB rows: (level, x, y, z, w, l, h, theta); some reach over the border (zero outside), some are smaller than one feature voxel
        obb = []
        for k in range(2):
            r = rand_obb(24, g, 4, 60, 3, 40)
            r[:, 1] = r[:, 1].clamp_max(60)
            r[:, 2] = r[:, 2].clamp_max(44)
            r[:4, 3:6] = torch.rand(4, 3, generator=g) * 3 + 0.5
            obb.append(torch.cat([torch.randint(0, 3, (24, 1), generator=g).float(), r], dim=1))
        for kind in ("pooling", "interpolation"):
            pool = R_det.ROIPool([3, 3, 3], scales, enlarge_scale=0.2, is_rotated_bbox=True, feature_extracting_type=kind)
            rois = torch.stack([r.clone() for r in obb])
            res = pool(feats, rois)
            out["obb_" + kind] = torch.stack(res)
            out["obb_rois_after_" + kind] = rois
        # AABB rows: (level, x0, y0, z0, x1, y1, z1) inside the grid (the reference slices with python semantics: keep the crops valid)
        aabb = []
        for k in range(2):
            lo = torch.rand(20, 3, generator=g) * torch.tensor([40., 30., 20.]) + 2
            hi = lo + torch.rand(20, 3, generator=g) * torch.tensor([36., 30., 24.]) + 1
            aabb.append(torch.cat([torch.randint(0, 3, (20, 1), generator=g).float(), lo, hi], dim=1))
        pool = R_det.ROIPool([2, 2, 2], scales, enlarge_scale=0.2, is_rotated_bbox=False)
        out["aabb_pooling"] = torch.stack(pool(feats, [r.clone() for r in aabb]))
    finally:
        torch.Tensor.get_device = orig_get_device
    for k in range(2):
        for l in range(3):
            out[f"feat{k}_{l}"] = feats[k][l]
    save("roipool", obb_rois=torch.stack(obb), aabb_rois=torch.stack(aabb), scales=torch.tensor(scales), **out)


def gen_ngp():
    """scripts/proposals2ngp.py: proposals -> instant-ngp bounding boxes (format-only consumer of the proposal files)."""
    print("ngp export")
    import importlib.util, json
    spec = importlib.util.spec_from_file_location("ref_p2ngp", "/root/reference/nerf_rpn/scripts/proposals2ngp.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(9)
    feats = dict(resolution=np.array([160, 120, 64]), bbox_min=np.array([-1.5, -2.0, -0.5]), bbox_max=np.array([2.5, 1.0, 1.5]),
                 scale=np.float64(0.33), offset=np.array([0.5, 0.4, 0.6]))
    aabb = np.concatenate([rng.random((5, 3)) * 60, rng.random((5, 3)) * 40 + 70], axis=1)
    obb = np.concatenate([rng.random((5, 3)) * 100 + 10, rng.random((5, 3)) * 30 + 5, rng.random((5, 1)) * 3 - 1.5], axis=1)
    out = {}
    for mitsuba in (False, True):
        f = dict(feats, from_mitsuba=mitsuba)
        out[f"aabb_{int(mitsuba)}"] = ref.proposals_to_ngp_boxes(aabb, f)
        out[f"obb_{int(mitsuba)}"] = ref.obb_to_ngp_boxes(obb, f)
    json.dump(dict(features={k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in feats.items()}, aabb=aabb.tolist(), obb=obb.tolist(),
                   boxes=out), open(os.path.join(HERE, "ngp_boxes.json"), "w"), indent=1)
    print("   wrote ngp_boxes.json")


def raw_scene_wlh4(shape, seed):
    """On-disk layout (W,L,H,4) f32: rgb U[0,1), density U[-5,5) (SURVEY 8d synthetic input for --normalize_density)."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.rand(*shape, 4, generator=g)
    raw[..., 3] = raw[..., 3] * 10.0 - 5.0
    return raw


def gen_fullsize():
    """BASELINE configs[1] at its real size (160^3, VGG19-EF, OBB, --normalize_density) and the reference's own benchmark shape
    200x200x130 (run_rpn.py:596).  Inputs are regenerated from seeds by the tests (a 160^3x4 grid is 65 MB); the fixture holds the
    expected outputs: sampled features, proposals, scores, levels."""
    print("full-size eval")
    import datasets as R_ds
    cases = [("eval_obb_160_cfg1", (160, 160, 160), True), ("eval_obb_200x200x130", (200, 200, 130), False)]
    only = os.environ.get("GOLDEN_ONLY")
    for name, shape, normalize in cases:
        if only and only not in name:
            continue
        ref = build_ref(True, 160).eval()
        orc = build_oracle(True, 160)
        orc.backbone.eval()
        raw = raw_scene_wlh4(shape, 300).numpy()
        if normalize:        # reference load_single_scene (datasets.py:39-63): alpha on channel 3, then (W,L,H,4) -> (4,W,L,H)
            raw = raw.copy()
            raw[..., 3] = R_ds.BaseDataset.density_to_alpha(raw[..., 3])
        x = torch.from_numpy(np.ascontiguousarray(np.transpose(raw, (3, 0, 1, 2)))).float()
        with torch.no_grad():
            (feats, props, lvls), _, scores = ref([x.clone()])
            (ofeats, oprops, olvls), _, oscores, aux = orc([x.clone()])
        arrs = {"shape": list(shape), "seed": 300, "normalize_density": normalize, "rotated": True, "resolution": 160, "pre": 2500}
        for i, (f, of) in enumerate(zip(feats, ofeats)):
            close(of, f, 5e-4, f"{name} feat{i}")
            idx, val = subsample(f, 16384)
            arrs[f"feat{i}_shape"], arrs[f"feat{i}_idx"], arrs[f"feat{i}_val"], arrs[f"feat{i}_absmax"] = list(f.shape), idx, val, f.abs().max()
        assert props[0].shape == oprops[0].shape, (name, props[0].shape, oprops[0].shape)
        close(oprops[0], props[0], 2e-3, f"{name} proposals"); close(oscores[0], scores[0], 1e-5, f"{name} scores")
        arrs["proposals0"], arrs["scores0"], arrs["levels0"] = props[0], scores[0], lvls[0]
        print(f"   {name}: {props[0].shape[0]} proposals, score range {scores[0].min():.4f}..{scores[0].max():.4f}")
        save(name, **arrs)


def gen_stages():
    """Integer stages of the eval path on the reference's REAL candidate distribution at BASELINE configs[1] size (160^3, VGG19-EF, OBB,
    --normalize_density; the scene of eval_obb_160_cfg1).  Hooks the reference (rpn.py:292-370, utils.py:215-265) and records: raw logits
    of all 950 625 anchors, the per-level top-k indices, the deltas / decoded boxes of those candidates, the candidates that reach NMS
    (after clip / small-box / score filters, quirk B3 included), batched_nms' keep indices and the final proposals."""
    print("stage-level eval at 160^3")
    import datasets as R_ds
    shape = (160, 160, 160)
    ref = build_ref(True, 160).eval()
    raw = raw_scene_wlh4(shape, 300).numpy().copy()
    raw[..., 3] = R_ds.BaseDataset.density_to_alpha(raw[..., 3])
    x = torch.from_numpy(np.ascontiguousarray(np.transpose(raw, (3, 0, 1, 2)))).float()
    rec = {}
    o_concat, o_topn, o_nms = R_rpn.concat_box_prediction_layers, R_rpn.RegionProposalNetwork._get_top_n_idx, R_rpn.batched_nms

    def concat(box_cls, box_reg, nd):
        out = o_concat(box_cls, box_reg, nd)
        rec["logits"], rec["deltas"] = out[0].detach().reshape(-1).clone(), out[1].detach().clone()
        return out

    def topn(self, objectness, num_anchors_per_level):
        r = o_topn(self, objectness, num_anchors_per_level)
        rec["topk_idx"], rec["per_level"] = r[0].clone(), list(num_anchors_per_level)
        return r

    def nms(boxes, scores, idxs, thr):
        keep = o_nms(boxes, scores, idxs, thr)
        rec["nms_boxes"], rec["nms_scores"], rec["nms_levels"], rec["nms_keep"] = boxes.clone(), scores.clone(), idxs.clone(), keep.clone()
        return keep
    R_rpn.concat_box_prediction_layers, R_rpn.RegionProposalNetwork._get_top_n_idx, R_rpn.batched_nms = concat, topn, nms
    o_decode = ref.rpn.box_coder.decode_list

    def decode_list(deltas_list, anchors_list):
        out = o_decode(deltas_list, anchors_list)
        rec["decoded"] = out.detach().clone()
        return out
    ref.rpn.box_coder.decode_list = decode_list
    try:
        with torch.no_grad():
            (feats, props, lvls), _, scores = ref([x.clone()])
    finally:
        R_rpn.concat_box_prediction_layers, R_rpn.RegionProposalNetwork._get_top_n_idx, R_rpn.batched_nms = o_concat, o_topn, o_nms
    idx = rec["topk_idx"]
    logits = rec["logits"]
    # tie report: equal logits inside a level's selected set / at its selection boundary make the index order implementation-defined (B7)
    ties, off, k0 = [], 0, 0
    for n in rec["per_level"]:
        k = min(2500, n)
        lv = logits[off:off + n]
        sel = lv[idx[k0:k0 + k] - off]
        srt = torch.sort(lv, descending=True).values
        ties.append([int((sel[1:] == sel[:-1]).sum()), int(k < n and srt[k - 1] == srt[k])])
        off += n; k0 += k
    dec = rec["decoded"][0]                       # [T, 8]: 7 box digits + level index
    # worst-case margin of the NMS decisions: |IoU - thr| over pairs of one level (kept row vs every lower-scored row)
    nb, nl = rec["nms_boxes"], rec["nms_levels"]
    margin = 1.0
    for l in torch.unique(nl):
        b = nb[nl == l]
        for i in range(0, b.shape[0], 256):
            iou = OB.iou_matrix(b[i:i + 256], b)
            margin = min(margin, float((iou - 0.3).abs().min()))
    print(f"   candidates {idx.numel()}, reach NMS {nb.shape[0]}, kept {rec['nms_keep'].numel()}, final {props[0].shape[0]}; ties {ties}; "
          f"min |IoU - 0.3| over same-level pairs {margin:.3e}")
    save("stages_obb_160_cfg1", shape=list(shape), seed=300, logits=logits, per_level=rec["per_level"], topk_idx=idx, ties=ties,
         topk_deltas=rec["deltas"][idx], topk_boxes=dec[idx, :7], nms_boxes=nb, nms_scores=rec["nms_scores"], nms_levels=nl,
         nms_keep=rec["nms_keep"], nms_margin=margin, proposals0=props[0], scores0=scores[0], levels0=lvls[0])


def gen_fullsize2():
    """Full-size goldens for the other two backbones (configs 2-4 use non-cubic grids 200x200x130 and 160x120x64):
    ResNet-50-3D + RPN, Swin-S-3D + RPN and Swin-S + FCOS, generated by running the reference here.  Inputs are regenerated from the
    seed by the tests; the fixtures hold sampled features and all proposals / scores."""
    print("full-size eval, ResNet-50 / Swin-S")
    only = os.environ.get("GOLDEN_ONLY")
    cases = [("eval_resnet_obb_200x200x130", (200, 200, 130), True, {"backbone": "resnet"}),
             ("eval_resnet_aabb_160x120x64", (160, 120, 64), False, {"backbone": "resnet"}),
             ("eval_swin_obb_160x120x64", (160, 120, 64), True, {"backbone": "swin"}),
             ("eval_swin_obb_200x200x130", (200, 200, 130), True, {"backbone": "swin"})]
    for name, shape, rot, kw in cases:
        if only and only not in name:
            continue
        ref = build_ref(rot, 160, **kw).eval()
        orc = build_oracle(rot, 160, **kw)
        orc.backbone.eval()
        x = scene(shape, 310)
        with torch.no_grad():
            (feats, props, lvls), _, scores = ref([x.clone()])
            (ofeats, oprops, olvls), _, oscores, aux = orc([x.clone()])
        arrs = {"shape": list(shape), "seed": 310, "rotated": rot, "resolution": 160, "pre": 2500, "backbone": kw["backbone"]}
        for i, (f, of) in enumerate(zip(feats, ofeats)):
            close(of, f, 5e-4, f"{name} feat{i}")
            idx, val = subsample(f, 16384)
            arrs[f"feat{i}_shape"], arrs[f"feat{i}_idx"], arrs[f"feat{i}_val"], arrs[f"feat{i}_absmax"] = list(f.shape), idx, val, f.abs().max()
        assert props[0].shape == oprops[0].shape, (name, props[0].shape, oprops[0].shape)
        close(oprops[0], props[0], 2e-3, f"{name} proposals"); close(oscores[0], scores[0], 1e-5, f"{name} scores")
        arrs["proposals0"], arrs["scores0"], arrs["levels0"] = props[0], scores[0], lvls[0]
        print(f"   {name}: {props[0].shape[0]} proposals, score range {scores[0].min():.4f}..{scores[0].max():.4f}")
        save(name, **arrs)
    fc = [("fcos_eval_obb_swin_200x200x130", (200, 200, 130), True, "swin"), ("fcos_eval_obb_swin_160x120x64", (160, 120, 64), True, "swin")]
    for name, shape, rot, bbk in fc:
        if only and only not in name:
            continue
        ref, orc = build_fcos(rot, bbk)
        ref.eval(); orc.backbone.eval()
        x = scene(shape, 320)
        with torch.no_grad():
            boxes, _, scores = ref([x.clone()])
            oboxes, _, oscores, aux = orc([x.clone()])
        arrs = {"shape": list(shape), "seed": 320, "shapes": [list(shape)], "rotated": rot, "backbone": bbk, "pre_nms_top_n": 2500,
                "fpn_post_nms_top_n": 2500, "pre_nms_thresh": 0.0, "min_size": 0.0}
        for key in ("box_cls", "box_reg", "centerness"):
            for l, t in enumerate(aux[key]):
                idx, val = subsample(t, 4096)
                arrs[f"{key}{l}_idx"], arrs[f"{key}{l}_val"], arrs[f"{key}{l}_absmax"] = idx, val, t.abs().max()
        assert boxes[0].shape == oboxes[0].shape, (name, boxes[0].shape, oboxes[0].shape)
        close(oboxes[0], boxes[0], 2e-3, f"{name} boxes"); close(oscores[0], scores[0], 1e-5, f"{name} scores")
        arrs["boxes0"], arrs["scores0"] = boxes[0], scores[0]
        print(f"   {name}: {boxes[0].shape[0]} boxes, score range {scores[0].min():.4f}..{scores[0].max():.4f}")
        save(name, **arrs)


def gen_headouts():
    """Raw RPN head outputs of the full-size ResNet-50 / Swin-S eval fixtures: the reference's objectness logits and box deltas
    BEFORE decode / top-k / NMS, so that the HIP forward can be held to north_star's literal 1e-4 on them (and the decoded-box errors of those
    cases tied to -- or separated from -- the delta errors).  Same scenes / weights as gen_fullsize2 (seed 310); own files, the fixtures
    of gen_fullsize2 are not rewritten.  Stored: 65 536 evenly spaced anchors' logits + deltas, and the logits / deltas of the reference's
    per-level top-k candidates (the rows decode reads)."""
    print("full-size head outputs, ResNet-50 / Swin-S")
    only = os.environ.get("GOLDEN_ONLY")
    cases = [("eval_resnet_obb_200x200x130", (200, 200, 130), True, {"backbone": "resnet"}),
             ("eval_swin_obb_160x120x64", (160, 120, 64), True, {"backbone": "swin"}),
             ("eval_swin_obb_200x200x130", (200, 200, 130), True, {"backbone": "swin"})]
    for name, shape, rot, kw in cases:
        if only and only not in name:
            continue
        ref = build_ref(rot, 160, **kw).eval()
        x = scene(shape, 310)
        rec = {}
        o_concat, o_topn = R_rpn.concat_box_prediction_layers, R_rpn.RegionProposalNetwork._get_top_n_idx

        def concat(box_cls, box_reg, nd):
            out = o_concat(box_cls, box_reg, nd)
            rec["logits"], rec["deltas"] = out[0].detach().reshape(-1).clone(), out[1].detach().clone()
            return out

        def topn(self, objectness, num_anchors_per_level):
            r = o_topn(self, objectness, num_anchors_per_level)
            rec["topk_idx"], rec["per_level"] = r[0].clone(), list(num_anchors_per_level)
            return r
        R_rpn.concat_box_prediction_layers, R_rpn.RegionProposalNetwork._get_top_n_idx = concat, topn
        try:
            with torch.no_grad():
                (feats, props, lvls), _, scores = ref([x.clone()])
        finally:
            R_rpn.concat_box_prediction_layers, R_rpn.RegionProposalNetwork._get_top_n_idx = o_concat, o_topn
        # the run must be the one the proposal fixture was taken from
        old = dict(np.load(os.path.join(HERE, name + ".npz")))
        assert np.array_equal(old["proposals0"], props[0].numpy()) and np.array_equal(old["scores0"], scores[0].numpy()), name
        logits, deltas, idx = rec["logits"], rec["deltas"].reshape(logits_n := rec["logits"].numel(), -1), rec["topk_idx"].reshape(-1)
        sidx, sval = subsample(logits, 65536)
        print(f"   {name}: {logits_n} anchors, |logit| max {logits.abs().max():.4f}, |delta| max {deltas.abs().max():.4f}, {idx.numel()} candidates")
        save("headout_" + name, shape=list(shape), seed=310, per_level=rec["per_level"], sample_idx=sidx, sample_logits=sval,
             sample_deltas=deltas[sidx], topk_idx=idx, topk_logits=logits[idx], topk_deltas=deltas[idx],
             logit_absmax=logits.abs().max(), delta_absmax=deltas.abs().max())


def gen_train():
    print("end-to-end train")
    cases = [("train_obb_160_cfg1", True, "smooth_l1", [(160, 160, 160)]),   # BASELINE configs[1] at its full size: the bench workload
             ("train_resnet_obb_iou_160x120x64", True, "iou", [(160, 120, 64)]),   # configs[4] family at a SURVEY 8d size
             ("train_swin_obb", True, "smooth_l1", [(80, 56, 48)]),      # stochastic depth 0 (the draw is RNG-stream specific)
             ("train_resnet_aabb", False, "smooth_l1", [(64, 56, 48)]),
             ("train_resnet_obb_iou", True, "iou", [(64, 56, 48)]),        # BASELINE configs[4]: ResNet-50 + rotated-IoU loss
             ("train_aabb", False, "smooth_l1", [(48, 48, 48)]),
             ("train_obb", True, "smooth_l1", [(48, 40, 32)]),
             ("train_obb_iou", True, "iou", [(48, 40, 32)]),
             ("train_obb_giou", True, "giou", [(48, 40, 32)]),
             ("train_obb_diou", True, "diou", [(48, 40, 32)]),
             ("train_aabb_batch2", False, "smooth_l1", [(48, 48, 32), (40, 32, 32)]),
             ("train_aabb_batch2_emptygt", False, "smooth_l1", [(48, 40, 32), (40, 32, 32)])]   # second scene has no GT boxes (the
             # reference's OBB path crashes on an empty scene: base_bbox_coder.py:16 concatenates 7- and 6-column targets)
    only = os.environ.get("GOLDEN_ONLY")
    for name, rot, loss, shapes in cases:
        if only and only not in name:
            continue
        bk = {"backbone": "resnet"} if "resnet" in name else ({"backbone": "swin", "sd": 0.0} if "swin" in name else {})
        ref = build_ref(rot, 160, loss, **bk).train()
        orc = build_oracle(rot, 160, loss, **bk)
        orc.backbone.train(); orc.rpn.head.train()
        xs = [scene(s, 200 + i) for i, s in enumerate(shapes)]
        g = torch.Generator().manual_seed(77)
        gts = []
        for s in shapes:
            nbox, smax = (16, 48) if min(s) >= 64 else (5, 20)      # full-size cases: the bench's 16 boxes of up to 48 voxels
            if rot:
                gt = rand_obb(nbox, g, 8, min(s) - 8, 6, smax)
            else:
                gt = rand_aabb(nbox, g, 8, min(s) - 8, 6, smax)
            gts.append(gt)
        if "emptygt" in name:
            gts[-1] = gts[-1][:0]
        torch.manual_seed(1234)
        _, losses, _ = ref([x.clone() for x in xs], [t.clone() for t in gts])
        total = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]
        total.backward()
        torch.manual_seed(1234)
        _, olosses, _, aux = orc([x.clone() for x in xs], [t.clone() for t in gts], training=True)
        ototal = olosses["loss_objectness"] + 5.0 * olosses["loss_rpn_box_reg"] + 0.0 * olosses["loss_rpn_box_reg_2d"]
        ototal.backward()
        arrs = {"shapes": shapes, "rotated": rot, "reg_loss_type": loss, "seed": 1234, "backbone": bk.get("backbone", "vgg")}
        for k in losses:
            e = close(olosses[k], losses[k], 2e-5 * max(1.0, abs(losses[k].item())), f"{name} {k}")
            arrs[k] = losses[k]
        arrs["pos_idx"], arrs["neg_idx"] = aux["sampled"]["pos"], aux["sampled"]["neg"]
        # the same algorithm evaluated in float64 (oracle == reference in fp32, so this is the reference algorithm's exact
        # value): lets the GPU tests judge cancellation-prone gradients (BN biases, stem weights) against the truth and
        # against the fp32 reference's own rounding error instead of against an arbitrary tolerance.
        o64 = build_oracle(rot, 160, loss, **bk)
        o64.backbone.double().train(); o64.rpn.head.double()
        ppos, pneg = aux["sampled"]["pos"], aux["sampled"]["neg"]
        o64.rpn.sampler_hook = lambda labels: (ppos, pneg)
        _, l64, _, _ = o64([x.double() for x in xs], [t.double() for t in gts], training=True)
        (l64["loss_objectness"] + 5.0 * l64["loss_rpn_box_reg"] + 0.0 * l64["loss_rpn_box_reg_2d"]).backward()
        p64 = dict(o64.backbone.named_parameters()); p64.update({"head." + k: v for k, v in o64.rpn.head.named_parameters()})
        arrs["labels"] = torch.cat(aux["labels"]).to(torch.int8)
        rp = dict(ref.backbone.named_parameters()); rp.update({"head." + k: v for k, v in ref.rpn.head.named_parameters()})
        op = dict(orc.backbone.named_parameters()); op.update({"head." + k: v for k, v in orc.rpn.head.named_parameters()})
        worst = record_grads(arrs, rp, op, p64)
        assert worst < 2e-3, (name, worst)
        for i, t in enumerate(gts):
            arrs[f"gt{i}"] = t
        print(f"   {name}: losses", {k: round(v.item(), 6) for k, v in losses.items()}, "pos", len(arrs["pos_idx"]),
              "worst rel grad err oracle-vs-ref", f"{worst:.2e}")
        save(name, **arrs)


def fcos_args(rot, **kw):
    """argparse defaults of run_fcos.py:30-131 with the flags of train_fcos.sh / test_fcos.sh."""
    import argparse
    a = dict(num_convs=4, norm_reg_targets=True, centerness_on_reg=True, rotated_bbox=rot, pre_nms_thresh=0.0, pre_nms_top_n=2500,
             nms_thresh=0.3, fpn_post_nms_top_n=2500, min_size=0.0, center_sampling_radius=1.5, iou_loss_type="iou",
             use_additional_l1_loss=False, proj2d_loss_weight=0.0)
    a.update(kw)
    return argparse.Namespace(**a)


def build_fcos(rot, backbone, **kw):
    """(reference FCOSOverNeRF, oracle FCOS) with identical seeded weights."""
    from model.fcos.fcos import FCOSOverNeRF
    from oracle import fcos as OF
    args = fcos_args(rot, **kw)
    if backbone == "swin":
        rb = SwinTransformer_FPN(patch_size=[4, 4, 4], window_size=[4, 4, 4], stochastic_depth_prob=0, expand_dim=True, **SWIN_S)
        ob = ON.SwinFPN(SWIN_S["embed_dim"], SWIN_S["depths"], SWIN_S["num_heads"], 0.0)
    else:
        rb, ob = VGG_FPN("EF", 4, True, 160), ON.VGGFPN("EF", 4, 160)
    seeded_state(rb, 1); seeded_state(ob, 1)
    ref = FCOSOverNeRF(args, rb, [4, 8, 16, 32])
    seeded_state(ref.fcos_module.head, 2, bias_jitter=0.5)
    for l, sc in enumerate(ref.fcos_module.head.scales):
        sc.scale.data.fill_(0.8 + 0.15 * l)
    oh = OF.FCOSHead(256, args.num_convs, [4, 8, 16, 32], args.norm_reg_targets, args.centerness_on_reg, rot)
    oh.load_state_dict(ref.fcos_module.head.state_dict())
    orc = OF.FCOS(ob, oh, [4, 8, 16, 32], rot, args.center_sampling_radius, args.iou_loss_type, args.norm_reg_targets,
                  args.use_additional_l1_loss, args.proj2d_loss_weight, args.pre_nms_thresh, args.pre_nms_top_n, args.nms_thresh,
                  args.fpn_post_nms_top_n, args.min_size)
    return ref, orc


def gen_fcos():
    print("FCOS end-to-end")
    only = os.environ.get("GOLDEN_ONLY")
    evals = [("fcos_eval_aabb_vgg", False, "vgg", [(64, 56, 48)], {}),
             ("fcos_eval_obb_swin", True, "swin", [(80, 56, 48)], {}),
             ("fcos_eval_obb_batch2", True, "vgg", [(64, 48, 48), (48, 40, 32)], {"pre_nms_top_n": 300, "fpn_post_nms_top_n": 400}),
             # candidate threshold on sigmoid(cls) and the small-box filter (inference.py:76-78, 134-136)
             ("fcos_eval_aabb_thresh", False, "vgg", [(64, 56, 48)], {"pre_nms_thresh": 0.2, "min_size": 4.0})]
    for name, rot, bbk, shapes, kw in evals:
        if only and only not in name:
            continue
        ref, orc = build_fcos(rot, bbk, **kw)
        ref.eval(); orc.backbone.eval()
        xs = [scene(s, 300 + i) for i, s in enumerate(shapes)]
        with torch.no_grad():
            boxes, _, scores = ref([x.clone() for x in xs])
            oboxes, _, oscores, aux = orc([x.clone() for x in xs])
        arrs = {"shapes": shapes, "rotated": rot, "backbone": bbk, "pre_nms_top_n": kw.get("pre_nms_top_n", 2500),
                "fpn_post_nms_top_n": kw.get("fpn_post_nms_top_n", 2500), "pre_nms_thresh": kw.get("pre_nms_thresh", 0.0),
                "min_size": kw.get("min_size", 0.0)}
        for key in ("box_cls", "box_reg", "centerness"):
            for l, t in enumerate(aux[key]):
                idx, val = subsample(t, 1024)
                arrs[f"{key}{l}_idx"], arrs[f"{key}{l}_val"] = idx, val
        for i in range(len(xs)):
            assert boxes[i].shape == oboxes[i].shape, (name, boxes[i].shape, oboxes[i].shape)
            close(oboxes[i], boxes[i], 2e-3, f"{name} boxes"); close(oscores[i], scores[i], 1e-5, f"{name} scores")
            arrs[f"boxes{i}"], arrs[f"scores{i}"] = boxes[i], scores[i]
            print(f"   {name}[{i}]: {boxes[i].shape[0]} boxes, score range {scores[i].min():.4f}..{scores[i].max():.4f}, levels",
                  torch.bincount(boxes[i][:, 0].long(), minlength=4).tolist())
        save(name, **arrs)

    trains = [("fcos_train_aabb_vgg", False, "vgg", [(64, 56, 48)], {}),
              ("fcos_train_aabb_giou_batch2", False, "vgg", [(64, 48, 48), (48, 40, 32)], {"iou_loss_type": "giou"}),
              ("fcos_train_obb_swin", True, "swin", [(80, 56, 48)], {}),
              ("fcos_train_obb_l1_proj", True, "vgg", [(64, 56, 48)], {"iou_loss_type": "linear_iou", "use_additional_l1_loss": True,
                                                                     "proj2d_loss_weight": 0.5}),
              ("fcos_train_obb_diou", True, "vgg", [(64, 56, 48)], {"iou_loss_type": "diou"}),
              ("fcos_train_obb_smoothl1", True, "vgg", [(64, 56, 48)], {"iou_loss_type": "smooth_l1"})]
    for name, rot, bbk, shapes, kw in trains:
        if only and only not in name:
            continue
        ref, orc = build_fcos(rot, bbk, **kw)
        ref.train(); orc.backbone.train()
        xs = [scene(s, 400 + i) for i, s in enumerate(shapes)]
        g = torch.Generator().manual_seed(78)
        gts = [(rand_obb(6, g, 10, min(s) - 10, 6, 36) if rot else rand_aabb(6, g, 4, min(s) - 14, 6, 40)) for s in shapes]
        _, losses, _ = ref([x.clone() for x in xs], [t.clone() for t in gts])
        (losses["loss_cls"] + losses["loss_reg"] + losses["loss_centerness"]).backward()
        _, olosses, _, aux = orc([x.clone() for x in xs], [t.clone() for t in gts], training=True)
        (olosses["loss_cls"] + olosses["loss_reg"] + olosses["loss_centerness"]).backward()
        arrs = {"shapes": shapes, "rotated": rot, "backbone": bbk, "iou_loss_type": kw.get("iou_loss_type", "iou"),
                "use_additional_l1_loss": kw.get("use_additional_l1_loss", False), "proj2d_loss_weight": kw.get("proj2d_loss_weight", 0.0)}
        for k in losses:
            close(olosses[k], losses[k], 2e-5 * max(1.0, abs(losses[k].item())), f"{name} {k}")
            arrs[k] = losses[k]
        arrs["labels"], arrs["reg_targets"], arrs["pos"] = aux["labels"].to(torch.int8), aux["reg_targets"], aux["pos"]
        _, o64 = build_fcos(rot, bbk, **kw)
        o64.backbone.double().train(); o64.head.double()
        _, l64, _, _ = o64([x.double() for x in xs], [t.double() for t in gts], training=True)
        (l64["loss_cls"] + l64["loss_reg"] + l64["loss_centerness"]).backward()
        name_of = lambda bb, hd: {**dict(bb.named_parameters()), **{"head." + k: v for k, v in hd.named_parameters()}}
        rp, op, p64 = name_of(ref.backbone, ref.fcos_module.head), name_of(orc.backbone, orc.head), name_of(o64.backbone, o64.head)
        unused = [k for k, p in rp.items() if p.grad is None]
        for k in unused:
            rp.pop(k); op.pop(k); p64.pop(k)
        arrs["unused"] = ",".join(unused)
        worst = record_grads(arrs, rp, op, p64)
        assert worst < 2e-3, (name, worst)
        for i, t in enumerate(gts):
            arrs[f"gt{i}"] = t
        print(f"   {name}: losses", {k: round(v.item(), 6) for k, v in losses.items()}, "pos", int(aux["pos"].numel()),
              "worst rel grad err oracle-vs-ref", f"{worst:.2e}", "unused", unused)
        save(name, **arrs)


def gen_convergence():
    """40 optimiser steps of the REFERENCE's training loop (run_rpn.py:345-349, 373-395: AdamW lr 1e-4 / wd 0.01, OneCycleLR over the
    run, clip_grad_norm 0.1, loss = objectness + 5 x box regression) on 4 fixed synthetic 160^3 scenes (VGG19-EF + FPN + RPN, OBB, 16 boxes
    each; the bench workload), cycling through the scenes.  Stored: the three losses of every step and the anchors the reference's sampler
    drew (labels depend on anchors and ground truth only, so the HIP run can be fed the same draws through ``sampler_hook``)."""
    print("convergence: 40 reference training steps at 160^3")
    import time
    from torch.optim import AdamW
    from torch.optim.lr_scheduler import OneCycleLR
    steps, nscene = int(os.environ.get("GOLDEN_CONV_STEPS", 40)), 4
    ref = build_ref(True, 160).train()
    xs = [scene((160, 160, 160), 300 + i) for i in range(nscene)]
    g = torch.Generator().manual_seed(78)
    gts = [rand_obb(16, g, 8, 152, 6, 48) for _ in range(nscene)]
    drawn = {}
    orig = ref.rpn.fg_bg_sampler

    class Rec:
        def __call__(self, labels):
            pos, neg = orig(labels)
            drawn["pos"], drawn["neg"] = torch.where(torch.cat(pos, dim=0))[0], torch.where(torch.cat(neg, dim=0))[0]
            return pos, neg

    ref.rpn.fg_bg_sampler = Rec()
    opt = AdamW(ref.parameters(), lr=1e-4, weight_decay=0.01)
    sched = OneCycleLR(opt, max_lr=1e-4, total_steps=steps)
    torch.manual_seed(4321)
    losses, pos_all, neg_all, pos_off, neg_off, gnorm = [], [], [], [0], [0], []
    for it in range(steps):
        t0 = time.time()
        k = it % nscene
        _, ls, _ = ref([xs[k].clone()], [gts[k].clone()])
        total = ls["loss_objectness"] + 5.0 * ls["loss_rpn_box_reg"] + 0.0 * ls["loss_rpn_box_reg_2d"]
        total.backward()
        gn = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        opt.step(); sched.step(); opt.zero_grad()
        losses.append([ls["loss_objectness"].item(), ls["loss_rpn_box_reg"].item(), ls["loss_rpn_box_reg_2d"].item()])
        gnorm.append(float(gn))
        pos_all.append(drawn["pos"]); neg_all.append(drawn["neg"])
        pos_off.append(pos_off[-1] + drawn["pos"].numel()); neg_off.append(neg_off[-1] + drawn["neg"].numel())
        print(f"   step {it}: obj {losses[-1][0]:.6f} reg {losses[-1][1]:.6f} |g| {gnorm[-1]:.4f}  ({time.time() - t0:.1f} s)", flush=True)
    arrs = {"steps": steps, "scenes": nscene, "losses": np.asarray(losses, dtype=np.float64), "grad_norm": np.asarray(gnorm),
            "pos": torch.cat(pos_all), "neg": torch.cat(neg_all), "pos_off": np.asarray(pos_off), "neg_off": np.asarray(neg_off)}
    for i, t in enumerate(gts):
        arrs[f"gt{i}"] = t
    save("convergence_obb_160", **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["geometry", "anchors", "coders", "matcher", "nms", "metrics", "cli", "detector", "roipool", "ngp", "eval", "fullsize", "train", "fcos"]
    for w in which:
        globals()["gen_" + w]()

"""CPU: the C-ABI library builds, loads and exports every symbol include/nerfrpn.h declares; host-side logic."""
import itertools
import os
import subprocess

import pytest
import torch

from nerf_rpn_amd import lib


def test_library_exports_every_declared_symbol():
    if not os.path.exists(lib.SO_PATH):
        lib.build()
    names = lib.declared_symbols()
    assert len(names) >= 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.SO_PATH]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    L = lib.load()   # also binds argtypes for every prototype; AttributeError on mismatch
    assert L.nrpn_abi_version() == 3
    assert lib.query("anchor_table_words", 4, 13) == 2 + 32 + 6 * 4 * 13
    assert lib.query("pool_out_size", 40, 2, 2, 0, 1) == 20 and lib.query("pool_out_size", 5, 2, 2, 0, 1) == 3
    assert lib.query("pool_out_size", 80, 3, 2, 1, 0) == 40


def test_header_is_plain_c(tmp_path):
    """include/nerfrpn.h is the drop-in boundary: it must compile as C99 (no C++ or torch types in the signatures) and every prototype must
    be bindable by the ctypes parser of lib.py (pointers, sizes, scalars only)."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "nerfrpn.h"\nint main(void) { nrpn_conv_opts o; o.size = (int)sizeof o; return o.size > 0 ? 0 : 1; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    protos = lib._prototypes()
    assert set(protos) == set(lib.declared_symbols())            # every declared entry point has a parsed prototype


def test_process_wide_switches_live_in_the_tools_header_only():
    """no global state at the boundary.  The nrpn_set_* process defaults (A/B timing switches) are declared in
    include/nerfrpn_tools.h; include/nerfrpn.h -- the drop-in boundary -- offers per-call nrpn_conv_opts only."""
    public = set(lib.declared_symbols(tools=False))
    everything = set(lib.declared_symbols())
    assert not [n for n in public if n.startswith("nrpn_set_")]
    setters = sorted(n for n in everything - public)
    # (+ one stateless test helper: the host evaluation of the kernels' multiply-shift division)
    assert setters and all(n.startswith("nrpn_set_") or n == "nrpn_fastdiv_host" for n in setters), setters
    src = open(os.path.join(os.path.dirname(lib.HEADER), "nerfrpn_tools.h")).read()
    assert '#include "nerfrpn.h"' in src


def test_argument_errors_are_reported_not_fatal():
    with pytest.raises(lib.NrpnError, match="box_dim"):
        lib.call("iou3d_matrix_f32", 0, 0, 0, 1, 1, 5, 0)
    with pytest.raises(lib.NrpnError, match="ksize"):
        lib.call("conv3d_fwd", 1, 1, 0, 1, 1, 4, 4, 4, 64, 64, 64, 5, 0, 0, 0, 0, 0)


def test_product_path_has_no_cpu_fallback():
    from nerf_rpn_amd import ops
    with pytest.raises(lib.NrpnError, match="non-CUDA"):
        ops.iou3d_pair(torch.zeros(1, 7), torch.zeros(1, 7))


def test_anchor_table_and_ratio_order():
    from nerf_rpn_amd import ops
    from oracle import anchors as OA
    assert tuple(ops.unique_ratio_permutations(ops.ASPECT_RATIOS[0])) == OA.RATIO_ORDER
    t = ops.AnchorTable((160, 160, 160), [(40,) * 3, (20,) * 3, (10,) * 3, (5,) * 3], device="cpu")
    assert t.A == 13 and t.total == 950625 and t.offsets == [0, 832000, 936000, 949000, 950625]
    assert t.strides == [(4,) * 3, (8,) * 3, (16,) * 3, (32,) * 3]
    t2 = ops.AnchorTable((200, 200, 130), [(50, 50, 33)] * 4, device="cpu")
    assert t2.strides[0] == (4, 4, 3)   # floor stride, quirk B2
    for s in ops.ANCHOR_SIZES:
        assert torch.equal(ops.base_anchor_table(s, ops.ASPECT_RATIOS[0]), OA.base_anchors(s))


def test_state_dict_keys_match_reference_layout():
    from nerf_rpn_amd.model import VGG_FPN, RPNHead
    from oracle import nets
    for res in (160, 64):
        a, b = VGG_FPN("EF", 4, True, res), nets.VGGFPN("EF", 4, res)
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys()) and len(sa) == 135
        assert all(sa[k].shape == sb[k].shape for k in sa)
    from nerf_rpn_amd.model.feature_extractor import ResNet_FPN_256, Bottleneck
    a, b = ResNet_FPN_256(Bottleneck, [3, 4, 6, 3], input_dim=4, is_max_pool=True), nets.ResNetFPN()
    sa, sb = a.state_dict(), b.state_dict()
    assert len(sa) == 332 and {k: v.shape for k, v in sa.items()} == {k: v.shape for k, v in sb.items()}
    for rot in (False, True):
        a, b = RPNHead(256, 13, 4, rotate=rot), nets.RPNHead(256, 13, 4, rot)
        assert {k: v.shape for k, v in a.state_dict().items()} == {k: v.shape for k, v in b.state_dict().items()}


def test_swin_and_fcos_state_dict_keys_match_reference_layout():
    """Same key names / shapes / buffer values as the oracle modules, which load the reference's state dicts strictly
    (tests/golden/make_golden.py): checkpoints of the reference's default swin_s / FCOS configuration interchange."""
    import argparse
    import torch
    from nerf_rpn_amd.model.feature_extractor import SwinTransformer_FPN
    from nerf_rpn_amd.model.fcos import FCOSModule
    from oracle import fcos as OF, nets
    a = SwinTransformer_FPN([4, 4, 4], 96, [2, 2, 18, 2], [3, 6, 12, 24], [4, 4, 4])
    b = nets.SwinFPN(96, (2, 2, 18, 2), (3, 6, 12, 24), 0.1)
    sa, sb = a.state_dict(), b.state_dict()
    assert len(sa) == 365 and {k: v.shape for k, v in sa.items()} == {k: v.shape for k, v in sb.items()}
    k = "stages.2.5.attn.relative_position_index"
    assert sa[k].dtype == torch.int64 and torch.equal(sa[k], sb[k])
    for rot in (False, True):
        args = argparse.Namespace(num_convs=4, norm_reg_targets=True, centerness_on_reg=True, rotated_bbox=rot, pre_nms_thresh=0.0,
                                  pre_nms_top_n=2500, nms_thresh=0.3, fpn_post_nms_top_n=2500, min_size=0.0, center_sampling_radius=1.5,
                                  iou_loss_type="iou", use_additional_l1_loss=False, proj2d_loss_weight=0.0)
        mod = FCOSModule(args, 256, [4, 8, 16, 32])
        ref = OF.FCOSHead(256, 4, [4, 8, 16, 32], True, True, rot)
        got = {k: v.shape for k, v in mod.state_dict().items()}
        assert got == {"head." + k: v.shape for k, v in ref.state_dict().items()}       # checkpoint key 'fcos_state_dict' holds head.*


def test_host_side_geometry_helpers():
    """Pure host logic used around the kernels: flattened FCOS location bookkeeping, ragged voxel lists, scene stacking."""
    import torch
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.model import hip_nn
    geom = ops.FcosGeometry(2, [(10, 8, 6), (5, 4, 3), (3, 2, 2)], [4, 8, 16])
    assert geom.counts == [480, 60, 12] and geom.total == 2 * 552
    assert geom.segment_offsets == [0, 480, 960, 1020, 1080, 1092, 1104]            # (level, scene) segments, level-major
    feats = [torch.arange(2 * x * y * z * 3, dtype=torch.float32).view(2, x, y, z, 3) for (x, y, z) in ((4, 3, 2), (2, 2, 1), (1, 1, 1))]
    rag, segs = hip_nn.ragged_cat(feats)
    assert rag.shape == (1, 2 * (24 + 4 + 1), 1, 1, 3) and segs == [(4, 3, 2)] * 2 + [(2, 2, 1)] * 2 + [(1, 1, 1)] * 2
    back = hip_nn.ragged_split(rag, feats)
    assert all(torch.equal(a, b) for a, b in zip(back, feats))
    # channels-last-backed scenes keep their memory layout through the stack, plain ones stack as usual
    cl = [torch.rand(5, 4, 3, 4).permute(3, 0, 1, 2) for _ in range(2)]
    st = ops.stack_scenes(cl)
    assert st.shape == (2, 4, 5, 4, 3) and st.permute(0, 2, 3, 4, 1).is_contiguous() and torch.equal(st[1], cl[1])
    plain = [torch.rand(4, 5, 4, 3) for _ in range(2)]
    assert torch.equal(ops.stack_scenes(plain), torch.stack(plain))
    from nerf_rpn_amd.datasets import RawScene
    rs = RawScene(torch.zeros(7, 6, 5, 4), 1)
    assert tuple(rs.shape) == (4, 7, 6, 5)
    import pickle
    assert pickle.loads(pickle.dumps(rs)).alpha_mode == 1                              # travels through DataLoader workers


def test_rank_to_core_pinning_plan():
    """affinity.plan: every local rank gets its own contiguous share of the cores of its GPU's NUMA node; without topology the allowed cores
    are divided evenly; shares never overlap and never leave the allowed set."""
    from nerf_rpn_amd import affinity
    allowed = list(range(0, 128))
    numa = {0: 0, 1: 0, 2: 0, 3: 0, 4: 1, 5: 1, 6: 1, 7: 1}
    cpus = {0: list(range(0, 64)), 1: list(range(64, 128))}
    shares = [affinity.plan(r, 8, allowed=allowed, numa_of=numa.get, cpus_of=cpus.get) for r in range(8)]
    for r, (cores, node) in enumerate(shares):
        assert node == numa[r] and len(cores) == 16 and cores == list(range(cores[0], cores[0] + 16))
        assert set(cores) <= set(cpus[node])
    flat = [c for cores, _ in shares for c in cores]
    assert len(flat) == len(set(flat)) == 128
    # no topology: even split of what the process may use (a cgroup-limited container: 24 cores, 8 ranks)
    shares = [affinity.plan(r, 8, allowed=list(range(8, 32)), numa_of=lambda d: None, cpus_of=lambda n: None) for r in range(8)]
    assert [c for cores, _ in shares for c in cores] == list(range(8, 32)) and all(node is None for _, node in shares)
    # fewer cores than ranks: ranks share, nobody is left without a core
    shares = [affinity.plan(r, 8, allowed=[0, 1, 2], numa_of=lambda d: None, cpus_of=lambda n: None) for r in range(8)]
    assert all(len(cores) == 1 and cores[0] in (0, 1, 2) for cores, _ in shares)
    # a single rank is never pinned (bench.py's cpu_baseline leg uses every core)
    assert affinity.pin_rank(0, 1)["pinned"] is False
    # two independent 2-rank jobs on one host (physical GPUs 0-1 and 2-3 of node 0, each job sees its own as devices 0, 1): the shares
    # follow the GPU's place among the node's four GPUs, so the jobs do not start at the same core
    phys = {"A": {0: 0, 1: 1}, "B": {0: 2, 1: 3}}
    got = {}
    for job, m in phys.items():
        for r in range(2):
            got[(job, r)] = affinity.plan(r, 2, allowed=allowed, numa_of=lambda d: 0, cpus_of=cpus.get, slot_of=lambda d, m=m: (m[d], 4))[0]
    flat = [c for cores in got.values() for c in cores]
    assert len(flat) == len(set(flat)) == 64 and got[("B", 0)][0] == 32 and all(len(c) == 16 for c in got.values())
    # without the host-wide view (no sysfs): the job's own ranks divide the node, as before
    assert affinity.plan(1, 2, allowed=allowed, numa_of=lambda d: 0, cpus_of=cpus.get, slot_of=lambda d: None)[0] == list(range(32, 64))


def test_multiply_shift_division_of_the_kernels_is_exact():
    """Round 6: the implicit-GEMM loaders, the pools and the top-down add decompose voxel indices with a multiply-shift pair made on the host
    (csrc/common.h: m = ceil(2^(31+l) / d), q = mulhi(n, m) >> (l - 1)).  The library's host evaluation of exactly that arithmetic must equal
    n // d for every n < 2^31: small divisors, powers of two and their neighbours, random ones, and the n around multiples of d."""
    import random
    from nerf_rpn_amd import lib
    L = lib.load()
    rnd = random.Random(0)
    ds = list(range(1, 200)) + [2 ** k + e for k in range(2, 31) for e in (-1, 0, 1)] + [rnd.randrange(1, 2 ** 31) for _ in range(300)]
    for d in ds:
        if not 1 <= d < 2 ** 31:
            continue
        ns = [0, 1, 2 ** 31 - 1, 2 ** 31 - 2, 2 ** 30] + [rnd.randrange(0, 2 ** 31) for _ in range(40)]
        ns += [k * d + e for k in (1, 2, (2 ** 31 - 1) // d) for e in (-1, 0, 1) if 0 <= k * d + e < 2 ** 31]
        for n in ns:
            assert L.nrpn_fastdiv_host(n, d) == n // d, (n, d)
    assert L.nrpn_fastdiv_host(2 ** 31, 3) == -1 and L.nrpn_fastdiv_host(5, 0) == -1


def test_pinned_bucket_size_is_validated(monkeypatch):
    """NRPN_GRAD_BUCKET_MIB is a configuration input -- garbage raises a message that names it instead of an int() traceback."""
    from nerf_rpn_amd.engine import FlatTrainer
    monkeypatch.delenv("NRPN_GRAD_BUCKET_MIB", raising=False)
    assert FlatTrainer._pinned_bucket_mib() is None
    monkeypatch.setenv("NRPN_GRAD_BUCKET_MIB", "32")
    assert FlatTrainer._pinned_bucket_mib() == 32
    for bad in ("0", "-4", "64MiB", "1e3", "99999"):
        monkeypatch.setenv("NRPN_GRAD_BUCKET_MIB", bad)
        with pytest.raises(ValueError, match="NRPN_GRAD_BUCKET_MIB"):
            FlatTrainer._pinned_bucket_mib()


def test_swin_b_of_the_reference_cli_is_rejected_like_the_reference_rejects_it():
    """run_rpn.py:284 lists swin_b as embed_dim 128 with heads [3, 6, 12, 24]; 128 is not divisible by 3, and the reference's attention
    (feature_extractor.py:446, qkv.reshape(..., 3, heads, C // heads)) raises on the first forward.  Here construction fails with a message
    that says so (every --backbone_type has been compared with the reference once -- for swin_b the comparison is 'both fail')."""
    import pytest
    from nerf_rpn_amd.model.feature_extractor import SwinTransformer_FPN
    with pytest.raises(ValueError, match="not divisible"):
        SwinTransformer_FPN(patch_size=[4, 4, 4], embed_dim=128, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=[4, 4, 4],
                            stochastic_depth_prob=0.1, expand_dim=True)


def test_recorded_experiment_patches_still_apply():
    """tools/patches/*.patch are measured-but-not-merged experiments that profiles/ and DESIGN cite (with the A/B script that needs each): they
    must keep applying to the kernel sources they were cut from, or the record is not reproducible."""
    import glob
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    patches = sorted(glob.glob(os.path.join(root, "tools", "patches", "*.patch")))
    assert patches, "tools/patches is cited by profiles/README.md"
    if shutil.which("git") is None:
        pytest.skip("git not available")
    for p in patches:
        r = subprocess.run(["git", "apply", "--check", p], cwd=root, capture_output=True, text=True)
        assert r.returncode == 0, f"{os.path.basename(p)} no longer applies: {r.stderr[:400]}"

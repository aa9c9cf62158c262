#!/usr/bin/env python
"""Throughput bench of the RPN hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one full training pass of the hot path over one synthetic scene per rank: 160^3 x 4 rgb-sigma grid ->
VGG19-3D + FPN + RPN head forward, target assignment, sampled losses, backward, bucketed RCCL gradient all-reduce,
fused clip + AdamW.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line with the
BASELINE.json metric (scenes/sec), a `roofline` object for the dominant kernel and a `cpu_baseline` object (the oracle,
timed on the host cores on a bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GRID = 160
NUM_GT = 16
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}      # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md


def synthetic_scene(seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(4, GRID, GRID, GRID, generator=g)
    g1 = torch.Generator().manual_seed(seed + 1000)
    ctr = torch.rand(NUM_GT, 3, generator=g1) * 120 + 20
    size = torch.rand(NUM_GT, 3, generator=g1) * 40 + 8
    theta = (torch.rand(NUM_GT, 1, generator=g1) - 0.5) * math.pi
    return x.to(device), torch.cat([ctr, size, theta], dim=1).to(device)


def build_backbone(kind):
    from nerf_rpn_amd.model import VGG_FPN, ResNet_FPN_256, Bottleneck, SwinTransformer_FPN
    if kind == "vgg":
        return VGG_FPN("EF", 4, True, GRID)
    if kind == "resnet":
        return ResNet_FPN_256(Bottleneck, [3, 4, 6, 3], input_dim=4, is_max_pool=True)
    return SwinTransformer_FPN(patch_size=[4, 4, 4], embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=[4, 4, 4],
                               stochastic_depth_prob=0.1 if kind == "swin" else 0.0, expand_dim=True)


def build_fcos(dtype, device, backbone):
    """Secondary workload (not the BASELINE metric): train_fcos.sh configuration, Swin-S + FCOS head, OBB."""
    from nerf_rpn_amd.model.fcos import FCOSOverNeRF
    torch.manual_seed(0)
    a = argparse.Namespace(num_convs=4, norm_reg_targets=True, centerness_on_reg=True, rotated_bbox=True, pre_nms_thresh=0.0,
                           pre_nms_top_n=2500, nms_thresh=0.3, fpn_post_nms_top_n=2500, min_size=0.0, center_sampling_radius=1.5,
                           iou_loss_type="iou", use_additional_l1_loss=False, proj2d_loss_weight=0.0)
    return FCOSOverNeRF(a, build_backbone(backbone), [4, 8, 16, 32], compute_dtype=dtype).to(device).train()


def build_model(dtype, device, backbone="vgg"):
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.model import RPNHead, NeRFRegionProposalNetwork, AnchorGenerator3D
    torch.manual_seed(0)
    bb = build_backbone(backbone)
    hd = RPNHead(256, 13, 4, rotate=True)
    model = NeRFRegionProposalNetwork(bb, AnchorGenerator3D(ops.ANCHOR_SIZES, ops.ASPECT_RATIOS), hd, rpn_pre_nms_top_n_train=2500,
                                      rpn_pre_nms_top_n_test=2500, rpn_post_nms_top_n_train=2500, rpn_post_nms_top_n_test=2500,
                                      rpn_nms_thresh=0.3, rpn_fg_iou_thresh=0.35, rpn_bg_iou_thresh=0.2, rpn_batch_size_per_mesh=256,
                                      rpn_positive_fraction=0.5, rotated_bbox=True, reg_loss_type="smooth_l1", compute_dtype=dtype)
    model.rpn.loss_2d_requires_grad = False     # reg_loss_weight_2d = 0 (run_rpn.py default); the value is still computed
    return model.to(device).train()


class ConvProbe:
    """HIP-event timing of every conv launch on the stream it is enqueued on (torch's current stream == the stream handed to
    the C ABI).  Used for the `roofline` object: algorithmic FLOPs of a launch / its measured duration."""

    def __init__(self):
        from nerf_rpn_amd import lib
        self.lib = lib
        self.records = []
        self.enabled = False
        self._orig = lib.call

    def install(self):
        orig = self._orig

        def call(name, *args):
            if not self.enabled or name not in ("conv3d_fwd", "conv3d_wgrad"):
                return orig(name, *args)
            n_, gx_, gy_, gz_, cin_, _, wrows_, k_ = args[4:12]
            if 2.0 * n_ * gx_ * gy_ * gz_ * cin_ * wrows_ * (k_ ** 3) < 1e11:     # only the heavy launches are timed
                return orig(name, *args)
            a = torch.cuda.Event(enable_timing=True)
            b = torch.cuda.Event(enable_timing=True)
            a.record()
            rc = orig(name, *args)
            b.record()
            if name == "conv3d_fwd":
                n, gx, gy, gz, cin, cout, wrows, k = args[4:12]
            else:
                n, gx, gy, gz, cin, cout, wrows, k = args[4:12]
            flops = 2.0 * n * gx * gy * gz * cin * wrows * (k ** 3)
            self.records.append((name, (n * gx * gy * gz, cin, wrows, k), flops, a, b))
            return rc
        self.lib.call = call
        import nerf_rpn_amd.ops as ops
        ops.call = call

    def summary(self, dtype_name):
        by = {}
        for name, shape, flops, a, b in self.records:
            ms = a.elapsed_time(b)
            e = by.setdefault((name, shape), [0, 0.0, flops])
            e[0] += 1
            e[1] += ms
        if not by:
            return None, []
        rows = sorted(((k, v) for k, v in by.items()), key=lambda kv: -kv[1][1])
        (name, shape), (cnt, ms, flops) = rows[0]
        avg_ms = ms / cnt
        achieved = flops / (avg_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[dtype_name]
        total_ms = sum(v[1] for _, v in rows)
        total_fl = sum(v[0] * v[2] for _, v in rows)
        vox, cin_, cout_, _k = shape
        if name == "conv3d_fwd":     # tile selection of launch_conv (csrc/conv3d.hip)
            big = cout_ >= 256 and -(-vox // 256) * -(-cout_ // 256) >= 200
            kernel = "conv_igemm_big_kernel" if big else "conv_igemm_kernel"
        else:
            kernel = "conv_wgrad_big_kernel" if (cin_ >= 256 and cout_ >= 256) else "conv_wgrad_kernel"
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_conv_256x256_40c.json")
        if os.path.exists(pmc) and shape == (64000, 256, 256, 3):
            for kname, vals in json.load(open(pmc))["kernels"].items():
                if kname.startswith(kernel):
                    traffic = vals.get("l2_miss_bytes (TCC_MISS_sum*128)", vals.get("hbm_bytes_est (FETCH_SIZE*2*1024 + WRITE_SIZE*1024)"))
        roof = {"bound": "mfma", "kernel": kernel,
                "shape": {"voxels": shape[0], "cin": shape[1], "cout": shape[2], "k": shape[3]}, "launches": cnt,
                "avg_ms": round(avg_ms, 4), "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_note": "bytes beyond L2 (MALL+HBM) per launch = TCC_MISS_sum x 128 B from rocprofv3 --pmc "
                                "(profiles/r01_pmc_conv_256x256_40c.json); algorithmic bytes of this launch = 69 MB (x 32.8 + w 3.5 + y 32.8)",
                "timed_heavy_launches": {"tflops": round(total_fl / (total_ms * 1e-3) / 1e12, 2), "ms_per_step": None}}
        return roof, rows


def cpu_baseline():
    """The oracle (CPU restatement of the reference, torch fp32 on the host cores): one fwd+bwd of the same 160^3 scene."""
    from oracle import nets as ON, rpn as OR
    torch.manual_seed(0)
    bb, hd = ON.VGGFPN("EF", 4, GRID), ON.RPNHead(256, 13, 4, True)
    det = OR.Detector(bb, OR.RPN(hd, rotated=True))
    bb.train()
    x, gt = synthetic_scene(0, "cpu")
    t0 = time.time()
    _, losses, _, _ = det([x], [gt], training=True)
    (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]).backward()
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 5), "unit": "scenes/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 fwd+bwd of one {GRID}^3x4 scene (VGG19-EF+FPN+RPN, OBB, fp32, no optimiser step), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="disable per-launch HIP-event timing of the conv kernels")
    ap.add_argument("--model", default="vgg_rpn", choices=["vgg_rpn", "resnet_rpn", "swin_rpn", "swin_fcos", "vgg_fcos"],
                    help="vgg_rpn = the BASELINE.json metric (default); the others are secondary workloads for profiling")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from nerf_rpn_amd import lib
    from nerf_rpn_amd.engine import FlatTrainer
    lib.call("check_device", local)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    backbone, head = args.model.split("_")
    fcos = head == "fcos"
    model = build_fcos(dtype, dev, "swin0" if backbone == "swin" else backbone) if fcos else build_model(dtype, dev, backbone)
    trainer = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=args.steps + args.warmup + 1)
    x, gt = synthetic_scene(rank, dev)
    probe = ConvProbe()
    if not args.no_probe:
        probe.install()

    def step():
        _, losses, _ = model([x], [gt])
        if fcos:
            loss = losses["loss_cls"] + losses["loss_reg"] + losses["loss_centerness"]
        else:
            loss = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]
        loss.backward()
        trainer.step()
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    probe.enabled = not args.no_probe
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    probe.enabled = False
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    final_loss = loss.item()

    if rank == 0:
        roof, rows = probe.summary(args.dtype) if not args.no_probe else (None, [])
        if roof is not None:
            conv_ms = sum(v[1] for _, v in rows) / args.steps
            roof["timed_heavy_launches"]["ms_per_step"] = round(conv_ms, 3)
        out = {
            "metric": "scenes/sec (160^3x4 grids, VGG19-3D+FPN+RPN fwd+bwd)", "value": round(world * args.steps / elapsed, 4),
            "unit": "scenes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: one 160x160x160x4 rgb-sigma grid per GPU, VGG19-EF 3D + FPN + anchor RPN (OBB, 16 GT "
                                   "boxes), fwd+bwd+clip+AdamW, random-init weights", "scenes_per_gpu": 1, "parallelism": f"dp{world}"},
            "final_loss": round(final_loss, 5),
            "roofline": roof,
        }
        if args.model != "vgg_rpn":
            out["metric"] = f"scenes/sec (160^3x4 grids, {args.model} fwd+bwd) -- secondary workload, not the BASELINE metric"
            out["config"]["workload"] = f"{args.model}: one 160x160x160x4 grid per GPU, 16 OBB GT boxes, fwd+bwd+clip+AdamW, random-init weights"
        if world == 1 and not args.no_cpu_baseline and args.model == "vgg_rpn":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""GPU time of the conv launches of one VGG19+RPN training step, bucketed by voxel count (HIP events around every launch)."""
import collections, os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, ROOT)
import bench
from nerf_rpn_amd import lib, ops
from nerf_rpn_amd.engine import FlatTrainer
dev = torch.device("cuda:0")
model = bench.build_model(torch.bfloat16, dev)
tr = FlatTrainer(model, lr=1e-4, total_steps=100)
x, gt = bench.synthetic_scene(0, dev)
def step():
    _, losses, _ = model([x], [gt])
    (losses["loss_objectness"] + 5 * losses["loss_rpn_box_reg"]).backward()
    tr.step()
for _ in range(2): step()
rec = []
orig = lib.call
def call(name, *a):
    if name not in ("conv3d_fwd", "conv3d_wgrad", "unpack_conv_wgrad"):
        return orig(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(name, *a); e1.record()
    vox = a[4] * a[5] * a[6] * a[7] if name != "unpack_conv_wgrad" else -1
    rec.append((name, vox, e0, e1))
    return r
lib.call = call; ops.call = call
step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for n, v, a, b in rec:
    agg[(n, v)][0] += 1; agg[(n, v)][1] += a.elapsed_time(b)
for (n, v), (c, ms) in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(f"{n:18s} voxels {v:7d}: {c:3d} launches {ms:6.2f} ms  ({ms / c * 1e3:6.1f} us each)")

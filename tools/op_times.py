"""Per-C-ABI-entry GPU time of one training step (HIP events around every lib.call; adds ~10 us per call, use for ranking only)."""
import collections, os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import bench
from nerf_rpn_amd import lib, ops
model_name = sys.argv[1] if len(sys.argv) > 1 else "swin_rpn"
dev = torch.device("cuda:0")
backbone, head = model_name.split("_")
fcos = head == "fcos"
model = bench.build_fcos(torch.bfloat16, dev, "swin0" if backbone == "swin" else backbone) if fcos else bench.build_model(torch.bfloat16, dev, backbone)
from nerf_rpn_amd.engine import FlatTrainer
tr = FlatTrainer(model, lr=1e-4, total_steps=100)
x, gt = bench.synthetic_scene(0, dev)
def step():
    _, losses, _ = model([x], [gt])
    (sum(losses.values()) if fcos else losses["loss_objectness"] + 5 * losses["loss_rpn_box_reg"]).backward()
    tr.step()
for _ in range(2): step()
rec = []
orig = lib.call
def call(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(name, *a); e1.record()
    rec.append((name, e0, e1))
    return r
lib.call = call; ops.call = call
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record(); step(); t1.record(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for n, a, b in rec:
    agg[n][0] += 1; agg[n][1] += a.elapsed_time(b)
tot = sum(v[1] for v in agg.values())
print(f"{model_name}: step {t0.elapsed_time(t1):.1f} ms (instrumented), inside C-ABI calls {tot:.1f} ms, {len(rec)} calls")
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {ms:7.2f} ms {c:5d} calls  {n}")

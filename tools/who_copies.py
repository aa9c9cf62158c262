"""Who issues the memcpy / memset calls of a training step: chrome trace of two steps (torch.profiler, python stacks), every runtime
memcpy/memset call attributed to the innermost repo-level python frame that encloses it in time.
    python tools/who_copies.py"""
import collections
import json
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from nerf_rpn_amd.engine import FlatTrainer  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(torch.bfloat16, dev, "vgg")
trainer = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=400)
x, gt = bench.synthetic_scene(0, dev)
gts = [gt.cpu()]


def step():
    _, losses, _ = model([x], gts)
    loss = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]
    loss.backward()
    trainer.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
path = os.path.join(tempfile.gettempdir(), "trace.json")
prof.export_chrome_trace(path)
ev = json.load(open(path))["traceEvents"]
py = [e for e in ev if e.get("cat") == "python_function" and "ts" in e and "dur" in e]
rt = [e for e in ev if e.get("cat") in ("cuda_runtime", "cuda_driver") and ("emcpy" in e.get("name", "") or "emset" in e.get("name", ""))]
print("runtime memcpy/memset calls in 2 steps:", len(rt), collections.Counter(e["name"] for e in rt))
agg = collections.Counter()
for r in rt:
    best = None
    for p in py:
        if p["tid"] == r["tid"] and p["ts"] <= r["ts"] and p["ts"] + p["dur"] >= r["ts"] + r.get("dur", 0):
            if "nerf_rpn_amd" in p["name"] or "bench" in p["name"] or "who_copies" in p["name"]:
                if best is None or p["dur"] < best["dur"]:
                    best = p
    agg[(r["name"], best["name"] if best else "(no repo frame: autograd thread?)")] += 1
for (n, w), c in agg.most_common(40):
    print(f"{c:4d}  {n:28s} {w}")
ops_ = [e for e in ev if e.get("cat") == "cpu_op" and e.get("name") in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::item", "aten::_local_scalar_dense")]
agg2 = collections.Counter()
for r in ops_:
    best = None
    for p in py:
        if p["tid"] == r["tid"] and p["ts"] <= r["ts"] and p["ts"] + p["dur"] >= r["ts"] + r.get("dur", 0):
            if "nerf_rpn_amd" in p["name"]:
                if best is None or p["dur"] < best["dur"]:
                    best = p
    agg2[(r["name"], best["name"] if best else "(no repo frame)")] += 1
print("aten copy-type ops:")
for (n, w), c in agg2.most_common(40):
    print(f"{c:4d}  {n:28s} {w}")

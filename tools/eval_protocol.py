"""The reference's --mode benchmark protocol (run_rpn.py:594-617: eval forwards on randn(4, 200, 200, 130) incl. decode / top-k / NMS),
split by stage with HIP events on the current stream: backbone + FPN, RPN head, proposal post-processing.
    python tools/eval_protocol.py [iters] [X Y Z]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shape = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else [200, 200, 130]
dev = torch.device("cuda", 0)
model = bench.build_model(torch.bfloat16, dev).eval()
x = torch.randn(4, *shape, generator=torch.Generator().manual_seed(0)).to(dev)
rpn = model.rpn
marks = []
orig_backbone, orig_filter, orig_head = model.backbone.forward, rpn.filter_proposals, rpn.head.forward


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def timed(name, fn):
    def wrap(*a, **k):
        e0 = ev()
        r = fn(*a, **k)
        marks.append((name, e0, ev()))
        return r
    return wrap


model.backbone.forward = timed("backbone+fpn", orig_backbone)
rpn.head.forward = timed("head", orig_head)
rpn.filter_proposals = timed("postprocess (incl. the read-back wait)", orig_filter)
with torch.no_grad():
    for _ in range(3):
        model([x])
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    for _ in range(iters):
        e0 = ev()
        (_, props, _), _, _ = model([x])
        marks.append(("whole forward", e0, ev()))
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / iters
import collections
acc = collections.defaultdict(float)
for n, a, b in marks:
    acc[n] += a.elapsed_time(b)
print(f"shape {shape}: wall {wall:.2f} ms per forward, {int(props[0].shape[0])} proposals")
for n, v in acc.items():
    print(f"  {n:45s} {v / iters:8.3f} ms")

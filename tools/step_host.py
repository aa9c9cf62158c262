"""Host side of the training step of bench.py (vgg_rpn, bf16, one 160^3 scene): enqueue time per step against the synchronised step time,
a cProfile of the enqueue work, and who issues the device-to-device copies (torch.profiler, python stacks).
    python tools/step_host.py [out_dir]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/host"
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from nerf_rpn_amd.engine import FlatTrainer  # noqa: E402

model = bench.build_model(torch.bfloat16, dev, "vgg")
trainer = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=400)
x, gt = bench.synthetic_scene(0, dev)
gts = [gt.cpu()]


def step():
    _, losses, _ = model([x], gts)
    loss = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]
    loss.backward()
    trainer.step()
    return loss


for _ in range(10):
    step()
torch.cuda.synchronize()
# (1) enqueue time of a step when the GPU is NOT the limiter: synchronise before each step, time until step() returns
enq = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    enq.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    step()
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / 30
enq.sort()
print(f"step {full * 1e3:.2f} ms; host enqueue per step (GPU idle at start): median {enq[len(enq) // 2] * 1e3:.2f} ms, min {enq[0] * 1e3:.2f} ms")
# (2) where the host time goes
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
open(os.path.join(out, "cprofile_tottime.txt"), "w").write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
open(os.path.join(out, "cprofile_cumulative.txt"), "w").write(s.getvalue())
# (3) device-to-device copies: who asks for them
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
rows = []
ev = prof.events()
by_corr = {}
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU:
        continue
    if "emcpy" in e.name or "copyBuffer" in e.name or "emset" in e.name:
        rows.append(e)
print("memcpy/memset device events:", len(rows))
import collections
agg = collections.Counter()
dur = collections.Counter()
for e in rows:
    # the launching CPU op: smallest CPU event whose time range contains the launch (linked through the correlation id when present)
    key = e.name
    agg[key] += 1
    dur[key] += e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
for k, v in agg.most_common():
    print(f"  {k[:80]:80s} {v:5d} {dur[k]:10.1f} us")
tab = prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=60, max_src_column_width=110)
open(os.path.join(out, "torch_profiler_by_stack.txt"), "w").write(tab)
cp = [l for l in tab.splitlines() if "copy" in l.lower() or "Memcpy" in l]
print("\n".join(l[:260] for l in cp[:40]))

"""Which torch ops issue device-to-device copies / fills in one training step of the bench configuration (torch.profiler, one step):
prints every op whose GPU kernels include a DtoD memcpy (`__amd_rocclr_copyBuffer`) or a fill, with the python call site."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerf_rpn_amd.engine import FlatTrainer  # noqa: E402

dev = torch.device('cuda:0')
model = bench.build_model(torch.bfloat16, dev)
tr = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=50)
x, gt = bench.synthetic_scene(0, dev)
gt = gt.cpu()


def step():
    _, losses, _ = model([x], [gt])
    (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]).backward()
    tr.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()
rows = {}
for e in evs:
    name = e.name
    if 'Memcpy' in name or 'copyBuffer' in name or 'Memset' in name or 'fillBuffer' in name or name in ('aten::copy_', 'aten::clone', 'aten::fill_', 'aten::zero_',
                                                                                                      'aten::contiguous', 'aten::cat', 'aten::stack'):
        stack = [s for s in (e.stack or []) if 'nerf_rpn_amd' in s or 'bench.py' in s][:2]
        key = (name, tuple(str(s) for s in (e.input_shapes or [])[:2]), tuple(stack))
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += e.device_time_total if hasattr(e, 'device_time_total') else 0.0
for (name, shapes, stack), (cnt, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{cnt:4d} x {name:28s} {us:9.1f} us  {shapes}  {" <- ".join(s.strip() for s in stack)}')

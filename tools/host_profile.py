"""cProfile of the host side of bench.py's training step (20 steps): where the Python / ctypes / allocator time per step goes."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from nerf_rpn_amd.engine import FlatTrainer  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
model = bench.build_model(torch.bfloat16, dev, 'vgg')
trainer = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=1000)
x, gt = bench.synthetic_scene(0, dev)


def step():
    _, losses, _ = model([x], [gt])
    loss = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]
    loss.backward()
    trainer.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 20
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime')
print(f'per step: total tottime {sum(v[2] for v in st.stats.values()) / N * 1e3:.2f} ms')
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:int(sys.argv[1]) if len(sys.argv) > 1 else 45]
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print(f'{tt / N * 1e6:8.1f} us/step  {nc / N:7.1f} calls  cum {ct / N * 1e6:8.1f}  {os.path.basename(fn)}:{line} {name}')

"""How much of the training step is the host?  Times bench.py's step (a) as is, (b) with the target preparation cached and the box
check skipped -- no host synchronisation inside the step --, (c) additionally with the 2-D projection loss stubbed; for (b)/(c) also the
host's own enqueue time per step.  (b) and (c) are NOT valid training steps; this is a diagnosis of where host work limits the GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
from nerf_rpn_amd.engine import FlatTrainer  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
model = bench.build_model(torch.bfloat16, dev, 'vgg')
trainer = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=1000)
x, gt = bench.synthetic_scene(0, dev)


gt_in = [gt]


def step():
    _, losses, _ = model([x], gt_in)
    loss = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]
    loss.backward()
    trainer.step()
    return loss


def run(name, n=30, warm=6):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f'{name}: {tot / n * 1e3:.2f} ms/step, host enqueue {host / n * 1e3:.2f} ms/step', flush=True)


run('a) as is (device-resident ground truth)')
gt_in[0] = gt.cpu()
run('a1) host ground truth: target preparation on its own stream')
from nerf_rpn_amd import ops  # noqa: E402
ops.set_wgrad_stream(True)
run('a2) wgrad on a side stream')
g1 = trainer.g_arena.clone() if hasattr(trainer, 'g_arena') else None
ops.set_wgrad_stream(False)
run('a) as is, again')
rpn = model.rpn
orig_prepare = rpn.prepare_targets
cache = {}


def cached_prepare(*a, **k):
    if 'p' not in cache:
        cache['p'] = orig_prepare(*a, **k)
    return cache['p']


rpn.prepare_targets = cached_prepare
model.check_bbox_degeneration = lambda targets: None
run('b) no sync, cached targets')
rpn._projection_loss = lambda pred, target, m: torch.zeros((), device=pred.device)
run('c) b + projection loss stubbed')

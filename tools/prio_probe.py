"""Does a stream priority help the training step?  The weight-gradient stream at lower priority / the main chain on a high-priority
stream, against the default (both priority 0).  python tools/prio_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from nerf_rpn_amd import ops  # noqa: E402
from nerf_rpn_amd.engine import FlatTrainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
try:
    print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
except Exception as e:  # noqa: BLE001
    print("no priority_range():", e)
model = bench.build_model(torch.bfloat16, dev, "vgg")
trainer = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=4000)
x, gt = bench.synthetic_scene(0, dev)
gts = [gt.cpu()]


def step():
    _, losses, _ = model([x], gts)
    loss = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]
    loss.backward()
    trainer.step()


def timed(tag, n=60):
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print(f"{tag:50s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", flush=True)


def reset_side(prio):
    torch.cuda.synchronize()
    ops._WGRAD_SIDE["streams"].clear()
    ops._WGRAD_SIDE["priority"] = prio


timed("default (both streams priority 0)")
for prio in (1, 2, -1):
    try:
        reset_side(prio)
        timed(f"weight-gradient stream priority {prio:+d}")
    except Exception as e:  # noqa: BLE001
        print("priority", prio, "failed:", e)
reset_side(0)
for hp in (-1,):
    hs = torch.cuda.Stream(device=dev, priority=hp)
    hs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(hs):
        timed(f"main chain on a priority {hp:+d} stream, wgrad stream 0")
    torch.cuda.current_stream().wait_stream(hs)
timed("default again")

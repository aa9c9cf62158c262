"""Clock and power of the GPU while the dominant conv launch runs back to back (rocm-smi sampled from a thread): random post-ReLU
operands against zeros, and the whole training step.  Evidence for "power-limited" in DESIGN.md.   python tools/power_probe.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nerf_rpn_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            out = str(e)
        samples.append((time.perf_counter(), out))
        time.sleep(0.25)


def parse(out):
    pw = re.search(r"Power[^\n]*?:\s*([0-9.]+)", out)
    sclk = re.search(r"sclk clock level[^\n]*\(([0-9.]+)Mhz\)", out)
    mclk = re.search(r"mclk clock level[^\n]*\(([0-9.]+)Mhz\)", out)
    return (float(pw.group(1)) if pw else None, float(sclk.group(1)) if sclk else None, float(mclk.group(1)) if mclk else None)


def window(tag, fn, seconds=6.0):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    b.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    vals = [parse(o) for t, o in samples if t0 + 1.0 <= t <= t1]
    pw = [v[0] for v in vals if v[0] is not None]
    sc = [v[1] for v in vals if v[1] is not None]
    print(f"{tag:46s} {a.elapsed_time(b) / n * 1e3:8.1f} us/iter   power {sum(pw) / max(1, len(pw)):7.1f} W ({len(pw)} samples)   "
          f"sclk {sum(sc) / max(1, len(sc)):7.1f} MHz (min {min(sc) if sc else 0:.0f}, max {max(sc) if sc else 0:.0f})", flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.5)
idle = [parse(o) for _, o in samples]
print("idle:", idle[-1], flush=True)
print(samples[-1][1][:1500], flush=True)
grid, cin, cout, k, dtype = 40, 256, 256, 3, torch.bfloat16
torch.manual_seed(0)
x = torch.randn(1, grid, grid, grid, cin, device=dev).clamp_min(0).to(dtype)
z = torch.zeros_like(x)
w = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
wp, _ = ops.PackedWeight().get([w], dtype, cout, False)
wz, _ = ops.PackedWeight().get([torch.zeros_like(w)], dtype, cout, False)
window("conv_halo 256->256@40^3, post-ReLU randn", lambda: ops._conv_fwd(x, wp, None, cout, cout, k, 0, dtype))
window("conv_halo 256->256@40^3, zeros", lambda: ops._conv_fwd(z, wz, None, cout, cout, k, 0, dtype))
window("conv_igemm_big (8 waves), post-ReLU randn", lambda: ops._conv_fwd(x, wp, None, cout, cout, k, 0, dtype, tile=lib.TILE_256X256))
import bench  # noqa: E402
from nerf_rpn_amd.engine import FlatTrainer  # noqa: E402
model = bench.build_model(torch.bfloat16, dev, "vgg")
trainer = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=100000)
xs, gt = bench.synthetic_scene(0, dev)
gts = [gt.cpu()]


def step():
    _, losses, _ = model([xs], gts)
    (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]).backward()
    trainer.step()


def steps():
    step()


for _ in range(5):
    step()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 8.0:
    for _ in range(20):
        step()
    n += 20
    torch.cuda.synchronize()
t1 = time.perf_counter()
vals = [parse(o) for t, o in samples if t0 + 1.0 <= t <= t1]
pw = [v[0] for v in vals if v[0] is not None]
sc = [v[1] for v in vals if v[1] is not None]
print(f"{'training step (VGG19+FPN+RPN, 160^3)':46s} {(t1 - t0) / n * 1e3:8.3f} ms/step  power {sum(pw) / max(1, len(pw)):7.1f} W   sclk {sum(sc) / max(1, len(sc)):7.1f} MHz "
      f"(min {min(sc) if sc else 0:.0f}, max {max(sc) if sc else 0:.0f})", flush=True)
stop = True

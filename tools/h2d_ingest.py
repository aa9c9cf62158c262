"""PCIe-inclusive ingest of one 160^3x4 scene: pinned host buffer in the on-disk layout -> device -> ingest kernel (bf16)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import ops
dev = torch.device('cuda:0')
for name, host in (("fp32", torch.rand(160, 160, 160, 4).pin_memory()), ("uint8", torch.randint(0, 256, (160, 160, 160, 4), dtype=torch.uint8).pin_memory())):
    for _ in range(3):
        ops.ingest_rgbsigma(host.to(dev, non_blocking=True), 0 if name == "uint8" else 1, torch.bfloat16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.ingest_rgbsigma(host.to(dev, non_blocking=True), 0 if name == "uint8" else 1, torch.bfloat16)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {host.numel() * host.element_size() / 1e6:.1f} MB  H2D + ingest {dt * 1e3:.2f} ms  ({host.numel() * host.element_size() / dt / 1e9:.1f} GB/s)")

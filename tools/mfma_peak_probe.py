"""What the MFMA array of this part sustains at its power cap: a register-only v_mfma_f32_32x32x16_bf16 loop (tools/mfma_peak_probe.hip: no
LDS, no VMEM in the loop; 8 accumulators per wave as in the production conv kernels) on zeros, on post-ReLU randn activations x small
randn weights (what the conv layers multiply), and on dense randn operands, with the shader clock and socket power sampled from rocm-smi.
Writes gpurun_out/r04_mfma_ceiling.json (copied to profiles/).     python tools/mfma_peak_probe.py [seconds per window]"""
import ctypes
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmfma_probe.so")
if not os.path.exists(SO):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "mfma_peak_probe.hip"), "-o", SO])
L = ctypes.CDLL(SO)
L.mfma_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.mfma_probe_launch.restype = ctypes.c_int

dev = torch.device("cuda:0")
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            out = str(e)
        samples.append((time.perf_counter(), out))
        time.sleep(0.2)


def parse(out):
    pw = re.search(r"Power[^\n]*?:\s*([0-9.]+)", out)
    sclk = re.search(r"sclk clock level[^\n]*\(([0-9.]+)Mhz\)", out)
    return (float(pw.group(1)) if pw else None, float(sclk.group(1)) if sclk else None)


def operands(kind):
    """[64 wave slots][12 fragments][64 lanes][8 bf16]: fragments 0-3 / 6-9 are A (activations), 4-5 / 10-11 are B (weights)."""
    g = torch.Generator(device="cpu").manual_seed(5)
    t = torch.randn(64, 12, 64, 8, generator=g)
    if kind == "zeros":
        t.zero_()
    elif kind == "relu_randn_x_w0.05":      # post-ReLU activations x N(0, 0.05) weights: the conv layers' operands
        t[:, [0, 1, 2, 3, 6, 7, 8, 9]] = t[:, [0, 1, 2, 3, 6, 7, 8, 9]].clamp_min(0)
        t[:, [4, 5, 10, 11]] *= 0.05
    elif kind == "relu_randn":
        t.clamp_min_(0)
    elif kind != "randn":
        raise ValueError(kind)
    return t.to(torch.bfloat16).to(dev).contiguous()


def window(kind, waves, seconds):
    ops = operands(kind)
    out = torch.zeros(256 * 512, dtype=torch.float32, device=dev)
    blocks, iters = 256, 4000
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        rc = L.mfma_probe_launch(ops.data_ptr(), out.data_ptr(), blocks, waves, iters, st)
        assert rc == 0, rc

    launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    n = 0
    a.record()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            launch()
        n += 50
        torch.cuda.synchronize()
    b.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = a.elapsed_time(b) / n
    flop = blocks * waves * iters * 16 * 32768.0
    vals = [parse(o) for t, o in samples if t0 + 1.0 <= t <= t1]
    pw = [v[0] for v in vals if v[0] is not None]
    sc = [v[1] for v in vals if v[1] is not None]
    res = {"operands": kind, "waves_per_simd": waves // 4, "ms_per_launch": round(ms, 4), "tflops": round(flop / ms / 1e9, 1),
           "frac_of_2500": round(flop / ms / 1e9 / 2500.0, 4), "power_w": round(sum(pw) / max(1, len(pw)), 1),
           "sclk_mhz": round(sum(sc) / max(1, len(sc)), 1), "sclk_min": min(sc) if sc else None, "samples": len(pw)}
    if sc:      # MFMA issue rate in cycles: 32 cycles per 32x32x16 MFMA per SIMD at full rate
        cyc = ms * 1e-3 * res["sclk_mhz"] * 1e6
        res["mfma_busy_in_cycles"] = round(iters * 16 * 32 * (waves // 4) / cyc, 4)
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(1.0)
    rows = []
    for waves in (8, 4):
        for kind in ("zeros", "relu_randn_x_w0.05", "relu_randn", "randn"):
            rows.append(window(kind, waves, seconds))
    stop = True
    os.makedirs(os.path.join(HERE, "..", "gpurun_out"), exist_ok=True)
    doc = {"what": "register-only v_mfma_f32_32x32x16_bf16 loop, 256 workgroups (one per CU), 8 accumulators per wave, two operand register sets "
                   "alternating; rocm-smi power / sclk averaged over the window (first second dropped)",
           "device": torch.cuda.get_device_name(0), "rows": rows}
    with open(os.path.join(HERE, "..", "gpurun_out", "r04_mfma_ceiling.json"), "w") as f:
        json.dump(doc, f, indent=1)

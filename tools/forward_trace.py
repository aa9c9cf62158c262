"""Eval-mode VGG19-3D + FPN forward on the bench scene, alone, for a kernel trace (a rocprofv3 table of the forward-only pass).

    rocprofv3 --kernel-trace --stats -d /tmp/fwd -o p --output-format csv -- python tools/forward_trace.py [iters]
    python tools/forward_trace.py --seq <kernel_trace.csv> [iters]      # launch sequence of the LAST forward: offset, duration, gap, grid

The run prints the HIP-event time per forward (the figure `bench.py` reports as `roofline.forward_vgg19_fpn.ms`) and the host's enqueue time.
"""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def seq(path, iters):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:70],
                     int(r.get("Grid_Size_X", 0) or 0) // max(1, int(r.get("Workgroup_Size_X", 1) or 1)), int(r.get("Workgroup_Size_X", 0) or 0)))
    rows.sort()
    # the trace holds warm-up + `iters` identical forwards; a forward starts at the layout conversion of the 4-channel scene
    marks = [i for i, r in enumerate(rows) if r[2].startswith("planes4_to_cl")]
    if len(marks) < 2:
        raise SystemExit("no forward boundaries found")
    per = marks[-1] - marks[-2]
    lo, hi = marks[-1], min(len(rows), marks[-1] + per)
    t0, end = rows[lo][0], rows[lo][0]
    busy = 0
    for s, e, n, wgs, wg in rows[lo:hi]:
        print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {max(0, s - end) / 1e3:5.1f}  wgs {wgs:6d} x {wg:4d}  {n}")
        busy += e - s
        end = max(end, e)
    print(f"forward span {(end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, {hi - lo} launches")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--seq":
        return seq(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20)
    import torch
    import bench
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    model = bench.build_model(torch.bfloat16, dev)
    x, _ = bench.synthetic_scene(0, dev)
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model.backbone(x.unsqueeze(0))
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        t0 = time.perf_counter()
        for _ in range(iters):
            model.backbone(x.unsqueeze(0))
        host = (time.perf_counter() - t0) / iters
        b.record()
        torch.cuda.synchronize()
    print(f"forward VGG19+FPN: {a.elapsed_time(b) / iters:.3f} ms per forward (HIP events), host enqueue {1e3 * host:.3f} ms")


if __name__ == "__main__":
    main()

"""Summary of a rocprofv3 --kernel-trace CSV of bench.py for profiles/: per-kernel totals (calls, total, average) and the launches of the
dominant shape on their own -- conv_halo_kernel (or conv_igemm_big_kernel) with a grid of 250 workgroups x 512 threads = 256->256 on 40^3 -- so that the
average can be compared with bench.py's HIP-event figure for the same shape.
    python tools/prof_summary.py <kernel_trace.csv> <out.json> [steps]"""
import collections
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
tot, cnt = collections.Counter(), collections.Counter()
dom, dom_name = [], "conv_halo_kernel"
for r in rows:
    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[name] += d
    cnt[name] += 1
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    if (name.startswith("conv_halo_kernel") or name.startswith("conv_igemm_big_kernel")) and wg == 512 and grid in (250 * 512, 250) and d > 120000:
        dom.append(d)       # 256->256 k3 on 40^3 (the 1^3 convs of that grid take ~25 us; 128->256 on the halo kernel ~100 us)
        dom_name = name
out = {"source": "rocprofv3 --kernel-trace", "launches": len(rows), "steps": steps,
       "gpu_time_ms_total": round(sum(tot.values()) / 1e6, 3),
       "dominant_shape": {"kernel": dom_name.split("<")[0], "shape": "256->256 k3 on 40^3 (250 workgroups)", "launches": len(dom),
                          "avg_us": round(sum(dom) / max(1, len(dom)) / 1e3, 2), "min_us": round(min(dom) / 1e3, 2) if dom else None,
                          "tflops_at_avg": round(226.4924 / (sum(dom) / max(1, len(dom)) / 1e9) / 1e3, 1) if dom else None},
       "kernels": [{"name": n, "calls": cnt[n], "total_ms": round(t / 1e6, 3), "avg_us": round(t / cnt[n] / 1e3, 2)} for n, t in tot.most_common(40)]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["dominant_shape"]), out["gpu_time_ms_total"])

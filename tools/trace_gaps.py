"""Idle time of the GPU inside the timed steps of a rocprofv3 --kernel-trace of bench.py (both streams): union of the kernel intervals
against the wall time of the steady-state window, the largest gaps with the kernels on either side, and the busy time per stream/queue.
    python tools/trace_gaps.py <kernel_trace.csv> [steps_to_keep=5] [out.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:60],
              r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows), key=lambda t: t[0])
# the optimiser kernel closes a step: keep the window between the (keep+1)-th last and the last adamw launches
ends = [e[1] for e in ev if e[2].startswith("adamw_kernel")]
if len(ends) < keep + 1:
    raise SystemExit(f"only {len(ends)} optimiser launches in the trace")
t0, t1 = ends[-keep - 1], ends[-1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, gaps, cur_end, last = 0, [], t0, "(previous step's adamw)"
per_q = {}
for s, e, n, q in win:
    per_q[q] = per_q.get(q, 0) + (e - s)
    if s > cur_end:
        gaps.append((s - cur_end, last, n))
        busy += e - s
        cur_end, last = e, n
    elif e > cur_end:
        busy += e - cur_end
        cur_end, last = e, n
wall = t1 - t0
gaps.sort(reverse=True)
out = {"steps": keep, "wall_ms_per_step": round(wall / keep / 1e6, 3), "busy_ms_per_step": round(busy / keep / 1e6, 3),
       "idle_ms_per_step": round((wall - busy) / keep / 1e6, 3), "idle_frac": round(1 - busy / wall, 4),
       "kernel_ms_per_step_by_queue": {q: round(v / keep / 1e6, 3) for q, v in per_q.items()},
       "gaps_over_5us_per_step": round(sum(1 for g in gaps if g[0] > 5000) / keep, 1),
       "idle_in_gaps_over_5us_ms_per_step": round(sum(g[0] for g in gaps if g[0] > 5000) / keep / 1e6, 3),
       "largest_gaps": [{"us": round(g[0] / 1e3, 1), "after": g[1], "before": g[2]} for g in gaps[:25]]}
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "largest_gaps"}))
for g in out["largest_gaps"][:25]:
    print(g)

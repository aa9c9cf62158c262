"""Timeline of ONE training step out of a rocprofv3 --kernel-trace CSV of bench.py (all streams): every launch with its queue, start offset
and duration, so that the critical path (what the main stream waits for, where a stream idles) can be read off.
    python tools/step_timeline.py <kernel_trace.csv> > timeline.txt"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:56],
              r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", 0) or 0) // max(1, int(r.get("Workgroup_Size_X", 1) or 1))) for r in rows))
ends = [e[1] for e in ev if e[2].startswith("adamw_kernel")]
t0, t1 = ends[-2], ends[-1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
queues = sorted({e[3] for e in win}, key=lambda q: -sum(e[1] - e[0] for e in win if e[3] == q))
col = {q: i for i, q in enumerate(queues)}
last_end = {q: t0 for q in queues}
print(f"step {1e-3 * (t1 - t0):.1f} us, {len(win)} launches, queues (by busy time): " +
      ", ".join(f"{q}: {1e-3 * sum(e[1] - e[0] for e in win if e[3] == q):.0f} us" for q in queues))
for s, e, n, q, wgs in win:
    gap = s - last_end[q]
    last_end[q] = e
    print(f"{1e-3 * (s - t0):9.1f}  q{col[q]}  dur {1e-3 * (e - s):7.1f}  idle-before {1e-3 * gap:7.1f}  wgs {wgs:6d}  {'    ' * col[q]}{n}")

"""Idle-gap analysis of a rocprofv3 --kernel-trace CSV: busy / idle time of the GPU inside the traced window and the kernels that most
often follow a gap.   python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps, gapn = collections.Counter(), collections.Counter()
end = rows[0][1]
for s, e, n in rows[1:]:
    if s > end:
        gaps[n] += s - end
        gapn[n] += 1
    end = max(end, e)
idle = sum(gaps.values())
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy(sum) {busy/1e6:.2f} ms  idle {idle/1e6:.2f} ms ({100*idle/span:.1f} %)")
for n, g in gaps.most_common(30):
    print(f"  {g/1e3:9.1f} us idle before {gapn[n]:5d} x {n}")

"""Kernel sequence of ONE step out of a rocprofv3 --kernel-trace CSV (the last occurrence of the adamw kernel back to the one before it):
start offset, duration and idle gap before every launch.   python tools/trace_seq.py <kernel_trace.csv> > seq.txt"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:90]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
pairs = [(a + 1, b + 1) for a, b in zip(marks, marks[1:]) if b - a > 100]      # whole steps only (bench.py also times adamw on its own)
lo, hi = min(pairs, key=lambda ab: rows[ab[1] - 1][1] - rows[ab[0]][0])
t0, end = rows[lo][0], rows[lo - 1][1]
for s, e, n in rows[lo:hi]:
    gap = max(0, s - end)
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  {n}")
    end = max(end, e)
print(f"step span {(rows[hi - 1][1] - t0) / 1e3:.1f} us, {hi - lo} launches")

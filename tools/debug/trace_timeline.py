#!/usr/bin/env python3
"""Print the kernel timeline of a few steps from a rocprofv3 --kernel-trace CSV (start offsets, durations, queue):
   python3 tools/debug/trace_timeline.py <dir with *kernel_trace.csv> [first_kernel_substring] [n_steps]"""
import csv, glob, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "advance_step"
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(idx) < nsteps + 12:
    print("too few steps", len(idx)); sys.exit(0)
a, b = idx[-nsteps - 2], idx[-2]
t0 = rows[a][0]
for s, e, n, q in rows[a:b]:
    short = n.split("(")[0].replace("void ", "").replace("aae::", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q{q}  {short}")
print("steps:", [(rows[idx[i + 1]][0] - rows[idx[i]][0]) / 1e3 for i in range(len(idx) - 6, len(idx) - 1)])

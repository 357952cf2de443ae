#!/bin/bash
# Kernel timeline of two steps INSIDE the timed fit() loop of bench.py (rocprofv3 --kernel-trace): start offset, duration,
# queue, kernel - gaps between dependent launches show.  Usage on the GPU box: bash tools/debug/fit_timeline.sh [bench args]
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$root" && mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp AAE_BENCH_NO_PROF=1 && rocprofv3 --kernel-trace --output-format csv -d "$root/gpurun_out/tr_fit" -o run -- python3 "$root/bench.py" --no-cpu --no-extra --steps 60 --warmup 10 --repeats 1 "$@" > /dev/null 2>&1 )
python3 - gpurun_out/tr_fit <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "dec_crit" in r[2]]
a, b = idx[40] - 3, idx[42] - 3
t0 = rows[a][0]
prev_end = {}
for s, e, n, q in rows[a:b]:
    short = n.split("(")[0].replace("void ", "").replace("aae::", "")[:34]
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  gap {gap:6.1f}  q{q} {short}")
PY
rm -rf gpurun_out/tr_fit

#!/bin/bash
# rocprofv3 kernel stats of predict -> rank calls (tools/rank_rate.py at RR_ROWS rows per call).  Usage on the GPU box:
#   bash tools/prof_rank.sh <tag> [rows]     -> gpurun_out/<tag>_rank_kernel_stats.csv
tag=${1:-r4}; rows=${2:-512}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root" && mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && RR_ROWS=$rows RR_DOCS=4096 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_prof_rank" -o run -- python3 "$root/tools/rank_rate.py" > "$root/gpurun_out/${tag}_prof_rank.log" 2>&1 )
find gpurun_out/${tag}_prof_rank -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_rank_kernel_stats.csv \;
rm -rf gpurun_out/${tag}_prof_rank
python3 - gpurun_out/${tag}_rank_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY

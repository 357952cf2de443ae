#!/bin/bash
# kernel stats of tools/vocab_rank_time.py <world> under rocprofv3 (per-rank compute of the vocabulary-sharded scheme)
w=${1:-8}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
VR_STEPS=100 VR_SCHEMES=${VR_SCHEMES:-vocab} rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/vr_prof -o run -- python3 $root/tools/vocab_rank_time.py $w > $root/gpurun_out/vr_prof.log 2>&1
cd $root
find gpurun_out/vr_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/vr_world${w}_kernel_stats.csv \;
rm -rf gpurun_out/vr_prof
grep "world" gpurun_out/vr_prof.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/vr_world${w}_kernel_stats.csv")))
for r in rows[:26]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
PY

#!/bin/bash
# rocprofv3 kernel stats of one rank's step of the dp_mode='shard' scheme at `world` ranks (collectives replaced by device
# copies: tools/vocab_rank_time.py).  Usage on the GPU box: bash tools/prof_shard.sh <tag> [world] [scheme]
tag=${1:-r4}; world=${2:-8}; scheme=${3:-shard}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root" && mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && VR_SCHEMES=$scheme VR_STEPS=100 VR_WARM=20 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_prof_shard" -o run -- python3 "$root/tools/vocab_rank_time.py" $world > "$root/gpurun_out/${tag}_prof_shard.log" 2>&1 )
find gpurun_out/${tag}_prof_shard -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_${scheme}_w${world}_kernel_stats.csv \;
rm -rf gpurun_out/${tag}_prof_shard
python3 - gpurun_out/${tag}_${scheme}_w${world}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  per step {float(r["TotalDurationNs"]) / 120 / 1e3:7.1f} us')
print("sum of kernel time per step: %.1f us" % (tot / 120 / 1e3))
PY

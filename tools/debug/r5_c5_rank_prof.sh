#!/bin/bash
# kernel stats of one rank's step at C5 (2.2 M items, 8 ranks, global batch 512, median length 60), schemes 'both' and 'shard'
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$root" && mkdir -p gpurun_out/r5
for scheme in both shard; do
  ( cd /tmp && export TMPDIR=/tmp && VR_N=2200000 VR_B=64 VR_MEDIAN_LEN=60 VR_BATCHES=8 VR_SCHEMES=$scheme VR_STEPS=20 VR_WARM=5 \
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/r5/c5prof_$scheme" -o run -- python3 "$root/tools/vocab_rank_time.py" 8 > "$root/gpurun_out/r5/c5prof_$scheme.log" 2>&1 )
  find gpurun_out/r5/c5prof_$scheme -name "*kernel_stats.csv" -exec cp {} gpurun_out/r5/c5_${scheme}_kernel_stats.csv \;
  rm -rf gpurun_out/r5/c5prof_$scheme
  echo "== $scheme"; grep "ms/step" gpurun_out/r5/c5prof_$scheme.log
  python3 - gpurun_out/r5/c5_${scheme}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  total {float(r["TotalDurationNs"]) / 1e3:9.1f} us')
PY
done

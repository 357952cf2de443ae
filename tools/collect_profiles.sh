#!/bin/bash
# Round-end evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag>
# Leaves under gpurun_out/: <tag>_bench.json (default bench.py run), <tag>_kernel_stats.csv (rocprofv3
# --kernel-trace --stats of the same command), pmc_fetch.txt / pmc_write.txt (two separate --pmc passes).
tag=${1:-r1}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root" && mkdir -p gpurun_out
python3 bench.py > gpurun_out/${tag}_bench.log 2>&1; tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_prof" -o run -- python3 "$root/bench.py" --no-cpu > "$root/gpurun_out/${tag}_prof.log" 2>&1 )
cp gpurun_out/${tag}_prof/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv 2>/dev/null || cp gpurun_out/${tag}_prof/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/${tag}_prof
bash tools/pmc_pass.sh fetch FETCH_SIZE
bash tools/pmc_pass.sh write WRITE_SIZE

#!/bin/bash
# rocprofv3 kernel stats of one rank's launches in a data-parallel scheme (run on the GPU box from the repo root):
#   bash tools/profile_dp.sh vocab|replicated   ->  gpurun_out/dp_<scheme>_kernel_stats.csv
scheme=${1:-vocab}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root" && mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/dp_prof" -o run -- python3 "$root/bench.py" --no-cpu --force-dp --dp $scheme --steps 200 > "$root/gpurun_out/dp_${scheme}_prof.log" 2>&1 )
cp gpurun_out/dp_prof/*kernel_stats.csv gpurun_out/dp_${scheme}_kernel_stats.csv 2>/dev/null || cp gpurun_out/dp_prof/*/*kernel_stats.csv gpurun_out/dp_${scheme}_kernel_stats.csv
rm -rf gpurun_out/dp_prof
tail -1 gpurun_out/dp_${scheme}_prof.log | cut -c1-200

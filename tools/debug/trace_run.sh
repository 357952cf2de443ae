#!/bin/bash
# rocprofv3 kernel trace of a short raw-step loop; usage: bash tools/debug/trace_run.sh <tag>   (env passes through)
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/trace_$tag -o run -- python3 $root/${SCRIPT:-tools/host_rate.py} > $root/gpurun_out/trace_$tag.log 2>&1
cd $root
python3 tools/debug/trace_timeline.py gpurun_out/trace_$tag ${ANCHOR:-advance_step} 1 > gpurun_out/timeline_$tag.txt
rm -rf gpurun_out/trace_$tag
tail -45 gpurun_out/timeline_$tag.txt

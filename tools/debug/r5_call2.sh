#!/bin/bash
# r5 call 2: full GPU suite on the new build, bench line with the new extras, dp_breakdown at world 1, C4 timeline again
set -u
o=gpurun_out/r5
mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests2.log 2>&1
echo "gpu tests rc=$?"; tail -3 $o/gpu_tests2.log
timeout -k 10 500 python bench.py --steps 100 --warmup 10 > $o/bench2.json 2> $o/bench2.err
echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --force-dp --no-cpu --no-extra > $o/bench2_dp1.json 2> $o/bench2_dp1.err
echo "bench dp1 rc=$?"
bash tools/debug/fit_timeline.sh --items 4587 --cond-inc 300 --batch 1000 > $o/tl_c4_2.txt 2>&1

#!/bin/bash
set -u
o=gpurun_out/r5
mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_parity_abi_gpu.py -m gpu -x -q -k "wide_batch_chain" > $o/x16_tests.log 2>&1
echo "x16 tests rc=$?"; tail -30 $o/x16_tests.log
root=$(pwd)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r5/x16_trace -o run -- python3 -m pytest $root/tests/test_parity_abi_gpu.py -m gpu -x -q -k "wide_batch_chain and step_cond_concat" > /dev/null 2>&1 )
find gpurun_out/r5/x16_trace -name "*kernel_stats.csv" | head -1 | xargs head -30
rm -rf gpurun_out/r5/x16_trace

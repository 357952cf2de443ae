#!/bin/bash
set -u
o=gpurun_out/r5; mkdir -p $o
python tools/debug/x16_vs_chain4.py 2>&1 | grep -v amdgpu.ids | tee $o/x16_vs_chain4.txt
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "beyond_two_gib" > $o/two_gib.log 2>&1
echo "2gib rc=$?"; tail -12 $o/two_gib.log

#!/bin/bash
set -u
o=gpurun_out/r5; mkdir -p $o
AAE_X16_ROWS=1 timeout -k 10 900 python -m pytest tests/test_parity_abi_gpu.py tests/test_fuzz_gpu.py tests/test_bf16_gpu.py tests/test_host_gpu.py -m gpu -q > $o/x16_forced2.log 2>&1
echo "forced rc=$?"; tail -6 $o/x16_forced2.log
for v in 1024 none; do
  if [ $v = none ]; then export AAE_NO_X16=1; unset AAE_X16_ROWS; else unset AAE_NO_X16; export AAE_X16_ROWS=$v; fi
  AAE_BENCH_EXTRAS=b512,c4 timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('x16 from $v: C3', d['ms_per_step'], ' b512', d['extra']['b512']['ms_per_step'], ' c4', d['extra']['c4']['ms_per_step'])"
done

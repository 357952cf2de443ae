#!/bin/bash
set -u
o=gpurun_out/r5; mkdir -p $o
AAE_DW_WIDE_ROWS=1 AAE_DW_WIDE_ANY=1 timeout -k 10 900 python -m pytest tests/test_parity_abi_gpu.py tests/test_fuzz_gpu.py tests/test_bf16_gpu.py tests/test_host_gpu.py -m gpu -q > $o/dw_wide_forced.log 2>&1
echo "forced rc=$?"; tail -5 $o/dw_wide_forced.log
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x > $o/dw_wide_fullsize.log 2>&1
echo "fullsize rc=$?"; tail -3 $o/dw_wide_fullsize.log
for rep in 1 2; do for v in wide narrow; do
  if [ $v = narrow ]; then export AAE_NO_DW_WIDE=1; else unset AAE_NO_DW_WIDE; fi
  AAE_BENCH_EXTRAS=b512,c4 python bench.py --no-cpu --steps 50 --warmup 10 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); e=d['extra']; print(os.environ['V'], 'C3', d['ms_per_step'], 'b512', e['b512']['ms_per_step'], 'c4', e['c4']['ms_per_step'], e['c4']['repeat_ms_per_step'])"
  VR_SCHEMES=shard VR_STEPS=100 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep -i "shard" | tail -1
done; done

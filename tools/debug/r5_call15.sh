#!/bin/bash
set -u
o=gpurun_out/r5; mkdir -p $o
AAE_X16_ROWS=1 timeout -k 10 900 python -m pytest tests/test_parity_abi_gpu.py tests/test_host_gpu.py tests/test_bf16_gpu.py -m gpu -q -x > $o/x16_la_forced.log 2>&1
echo "forced rc=$?"; tail -3 $o/x16_la_forced.log
CT_N=4587 CT_B=1000 CT_COND=300 AAE_X16_ROWS=256 AAE_CHAIN_TS=1 python tools/debug/chain_ts_shape.py 2> $o/cts16_la.txt; grep "chain16x3" $o/cts16_la.txt | tail -4
for v in 1024 512 256; do
  AAE_X16_ROWS=$v AAE_BENCH_EXTRAS=b512,c4 python bench.py --no-cpu --steps 50 --warmup 10 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); e=d['extra']; print('x16 from', os.environ['V'], 'C3', d['ms_per_step'], 'b512', e['b512']['ms_per_step'], 'c4', e['c4']['ms_per_step'])"
  AAE_X16_ROWS=$v VR_SCHEMES=shard VR_STEPS=100 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep -i "shard" | tail -1
done

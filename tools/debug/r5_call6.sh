#!/bin/bash
# same-box A/B: wide-batch chain kernel from 1024 rows on vs never
set -u
o=gpurun_out/r5; mkdir -p $o
for rep in 1 2; do
for v in 1024 none; do
  if [ $v = none ]; then export AAE_NO_X16=1; unset AAE_X16_ROWS; else unset AAE_NO_X16; export AAE_X16_ROWS=$v; fi
  AAE_BENCH_EXTRAS=b512,c4 timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('x16 from $v rep $rep: C3', d['ms_per_step'], ' b512', d['extra']['b512']['ms_per_step'], ' c4', d['extra']['c4']['ms_per_step'], d['extra']['c4']['repeat_ms_per_step'])"
  VR_SCHEMES=shard VR_STEPS=100 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep -i "shard" | tail -1
done
done

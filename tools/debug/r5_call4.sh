#!/bin/bash
# same-box A/B: wide-batch chain kernel (default) vs AAE_NO_X16=1
set -u
o=gpurun_out/r5
mkdir -p $o
for rep in 1 2; do
for v in x16 nox16; do
  if [ $v = nox16 ]; then export AAE_NO_X16=1; else unset AAE_NO_X16; fi
  AAE_BENCH_EXTRAS=b512,c4 timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v rep $rep: C3', d['ms_per_step'], ' b512', d['extra']['b512']['ms_per_step'], ' c4', d['extra']['c4']['ms_per_step'], d['extra']['c4']['repeat_ms_per_step'])"
  VR_SCHEMES=shard VR_STEPS=100 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep -i "shard" | tail -2
done
done
unset AAE_NO_X16
bash tools/debug/fit_timeline.sh --items 4587 --cond-inc 300 --batch 1000 > $o/tl_c4_x16.txt 2>&1
bash tools/debug/fit_timeline.sh --batch 512 > $o/tl_b512_x16.txt 2>&1

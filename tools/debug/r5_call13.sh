#!/bin/bash
set -u
for rep in 1 2 3; do for v in claim noclaim; do
  if [ $v = noclaim ]; then export AAE_NO_BUCKET_CLAIM=1; else unset AAE_NO_BUCKET_CLAIM; fi
  AAE_BENCH_EXTRAS=b512,c4 python bench.py --no-cpu --steps 50 --warmup 10 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); e=d['extra']; print(os.environ['V'], 'C3', d['ms_per_step'], 'b512', e['b512']['ms_per_step'], 'c4', e['c4']['ms_per_step'], e['c4']['repeat_ms_per_step'])"
  VR_SCHEMES=shard VR_STEPS=100 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep -i "shard" | tail -1
done; done

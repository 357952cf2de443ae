#!/bin/bash
set -u
for rep in 1 2 3; do
  AAE_BENCH_EXTRAS=b512,c4,c5_world1 python bench.py --no-cpu --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); e=d['extra']; print('C3', d['ms_per_step'], 'b512', e['b512']['ms_per_step'], 'c4', e['c4']['ms_per_step'], 'c5', e['c5_world1']['ms_per_step'])"
done
python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "batch_512 or c5_whole or c4_bench" 2>&1 | tail -2

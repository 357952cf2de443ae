#!/bin/bash
# grouped_dw_kernel's register allocation aimed at 4 | 5 | 6 waves per SIMD (128 | 96 | 80 registers: 4 | 5 | 6 workgroups per CU)
set -u
line() { AAE_BENCH_EXTRAS=b512,c4 python bench.py --no-cpu --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']; print('C3', d['ms_per_step'], 'b512', e['b512']['ms_per_step'], 'c4', e['c4']['ms_per_step'])"; }
probe() { VR_SCHEMES=shard VR_STEPS=150 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep "ms/step" | sed "s/^.*shard')//; s/of compute.*//"; }
for rep in 1 2; do
  echo "4 waves: $(line) | world 8: $(probe)"
  for w in 5 6; do echo "$w waves: $(AAE_HIP_LIB=$PWD/tools/debug/libaaerec_w$w.so line) | world 8: $(AAE_HIP_LIB=$PWD/tools/debug/libaaerec_w$w.so probe)"; done
done

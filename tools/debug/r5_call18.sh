#!/bin/bash
# the early prefetch (next batch's item list + catch-up beside the step's opening gather) after the gathers got twice as fast
set -u
line() { AAE_BENCH_EXTRAS=b512,c4 python bench.py --no-cpu --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']; print('C3', d['ms_per_step'], 'b512', e['b512']['ms_per_step'], 'c4', e['c4']['ms_per_step'])"; }
probe() { VR_SCHEMES=shard VR_STEPS=150 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep "ms/step" | sed "s/^.*shard')//; s/of compute.*//"; }
for rep in 1 2 3; do
  echo "default:            $(line) | world 8: $(probe)"
  echo "no early prefetch:  $(AAE_NO_EARLY_PREFETCH=1 line) | world 8: $(AAE_NO_EARLY_PREFETCH=1 probe)"
done

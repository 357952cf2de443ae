#!/bin/bash
# first-layer item workgroups of the weight-gradient launches: sized by the last distinct-item count seen | one per possible item
set -u
line() { AAE_BENCH_EXTRAS=b512,c4 python bench.py --no-cpu --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']; print('C3', d['ms_per_step'], 'b512', e['b512']['ms_per_step'], 'c4', e['c4']['ms_per_step'])"; }
probe() { VR_SCHEMES=shard VR_STEPS=150 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep "ms/step" | sed "s/^.*shard')//; s/of compute.*//"; }
for rep in 1 2; do
  echo "by the last count: $(line) | world 8: $(probe)"; echo "per possible item: $(AAE_NO_ITEM_COUNT=1 line) | world 8: $(AAE_NO_ITEM_COUNT=1 probe)"
done

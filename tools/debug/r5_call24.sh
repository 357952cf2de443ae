#!/bin/bash
# wide batches over a small vocabulary (C4, an item slice): the hybrid item form (four items per workgroup, many-row items by the whole
# workgroup behind a barrier) against one workgroup per item
set -u
c4() { AAE_BENCH_EXTRAS=c4 python bench.py --no-cpu --steps 30 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['extra']['c4']['ms_per_step'])"; }
probe() { VR_SCHEMES=shard VR_STEPS=150 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep "ms/step" | sed "s/^.*shard')//; s/of compute.*//"; }
for rep in 1 2; do
  echo "workgroup form: c4 $(c4) world 8 $(probe)"
  echo "hybrid form:    c4 $(AAE_W1_HYBRID_WIDE=1 c4) world 8 $(AAE_W1_HYBRID_WIDE=1 probe)"
done
AAE_W1_HYBRID_WIDE=1 AAE_DW_TS=60 python bench.py --no-cpu --no-extra --items 4587 --cond-inc 300 --batch 1000 --steps 30 --warmup 10 2>&1 | grep -A3 "grouped_dw launch 6[02]" | cut -c1-210
AAE_W1_HYBRID_WIDE=1 python -m pytest tests -m gpu -q -x -k "c4 or shard" 2>&1 | tail -2

#!/bin/bash
# batches of one fused launch: the hybrid item form's workgroups sized by the last distinct-item count | by the bound (800 at C3)
set -u
line() { python bench.py --no-cpu --no-extra --steps 300 --warmup 30 $* 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "C3 by the count: $(AAE_ITEM_COUNT_HYBRID=1 line) | by the bound: $(line) | 512 workgroups: $(AAE_W1_WGS=512 line) | 640: $(AAE_W1_WGS=640 line)"
done
echo "C2 bf16 by the count: $(AAE_ITEM_COUNT_HYBRID=1 line --dtype bf16) | by the bound: $(line --dtype bf16)"
echo "C2 bf16 by the count: $(AAE_ITEM_COUNT_HYBRID=1 line --dtype bf16) | by the bound: $(line --dtype bf16)"

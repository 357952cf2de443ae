#!/bin/bash
# weight-gradient launches: k-split form of a tile (rows >= 256) on | off, with and without the 16-wave kernel
set -u
line() { AAE_BENCH_EXTRAS=b512,c4 python bench.py --no-cpu --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']; print('C3', d['ms_per_step'], 'b512', e['b512']['ms_per_step'], 'c4', e['c4']['ms_per_step'])"; }
probe() { VR_SCHEMES=shard VR_STEPS=150 timeout -k 10 200 python tools/vocab_rank_time.py 8 2>/dev/null | grep "ms/step" | sed "s/^.*shard')//"; }
for rep in 1 2; do
  echo "k-split + 16-wave kernel: $(line) | world 8: $(probe)"
  echo "k-split only:             $(AAE_NO_DW_WIDE=1 line) | world 8: $(AAE_NO_DW_WIDE=1 probe)"
  echo "neither (r5 so far):      $(AAE_DW_KSPLIT_ROWS=0 line) | world 8: $(AAE_DW_KSPLIT_ROWS=0 probe)"
done
python -m pytest tests -m gpu -q -x -k "fullsize or fuzz or shard or wide or c4 or parallel" 2>&1 | tail -3

#!/bin/bash
# per-workgroup clocks of the weight-gradient launches (AAE_DW_TS), C4 / C3 at batch 512 / one rank's step at world 8 / C3
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$root" && mkdir -p gpurun_out/r5
echo "== C4"; AAE_DW_TS=60 python bench.py --no-cpu --no-extra --items 4587 --cond-inc 300 --batch 1000 --steps 30 --warmup 10 2>&1 | grep -A3 "grouped_dw launch"
echo "== C3 batch 512"; AAE_DW_TS=60 python bench.py --no-cpu --no-extra --batch 512 --steps 30 --warmup 10 2>&1 | grep -A3 "grouped_dw launch"
echo "== world 8"; AAE_DW_TS=90 VR_SCHEMES=shard VR_STEPS=40 VR_WARM=10 python tools/vocab_rank_time.py 8 2>&1 | grep -A3 "grouped_dw launch"
echo "== C3"; AAE_DW_TS=90 python bench.py --no-cpu --no-extra --steps 40 --warmup 10 2>&1 | grep -A3 "grouped_dw launch"

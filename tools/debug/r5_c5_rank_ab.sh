#!/bin/bash
# one rank's step at C5 (shard scheme) with this round's switches off one at a time, same box
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$root" && mkdir -p gpurun_out/r5
run() { VR_N=2200000 VR_B=64 VR_MEDIAN_LEN=60 VR_BATCHES=8 VR_SCHEMES=shard VR_STEPS=40 VR_WARM=5 timeout -k 10 300 python3 tools/vocab_rank_time.py 8 2>/dev/null | grep "ms/step" | sed "s/^.*shard')//"; }
for rep in 1 2; do
  echo "default: $(run)"
  echo "AAE_NO_X16: $(AAE_NO_X16=1 run)"
  echo "AAE_NO_BUCKET_CLAIM: $(AAE_NO_BUCKET_CLAIM=1 run)"
  echo "AAE_NO_DW_WIDE: $(AAE_NO_DW_WIDE=1 run)"
  for w in 160 192 208; do echo "AAE_OBK_WGS=$w: $(AAE_OBK_WGS=$w run)"; done
done

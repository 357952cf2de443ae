#!/bin/bash
# per-rank compute at world 2 / 4 / 8, 'both' (dp_mode='vocab') and 'shard', three probes each; the new rank test
set -u
o=gpurun_out/r5; mkdir -p $o
for w in 2 4 8; do for rep in 1 2 3; do
  VR_SCHEMES=both,shard VR_STEPS=150 timeout -k 10 200 python tools/vocab_rank_time.py $w 2>/dev/null | grep "ms/step" | sed "s/^/probe $rep: /"
done; done | tee $o/dp_rank_compute.log
timeout -k 10 300 python -m pytest tests/test_rank_gpu.py -m gpu -x -q -k "saturated" 2>&1 | tail -4

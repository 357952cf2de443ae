#!/bin/bash
set -u
o=gpurun_out/r5; mkdir -p $o
echo "== (a) x16 forced"; AAE_X16_ROWS=1 timeout -k 10 300 python -m pytest tests/test_host_gpu.py -m gpu -x -q -k "c3_scale_ranking" 2>&1 | grep -E "probe items|passed|failed"
echo "== (b) chain4, no piggyback"; AAE_NO_PIGGYBACK=1 timeout -k 10 300 python -m pytest tests/test_host_gpu.py -m gpu -x -q -k "c3_scale_ranking" 2>&1 | grep -E "probe items|passed|failed"
echo "== (c) x16 forced, no prefetch hints"; AAE_X16_ROWS=1 AAE_NO_EARLY_PREFETCH=1 timeout -k 10 300 python -m pytest tests/test_host_gpu.py -m gpu -x -q -k "c3_scale_ranking" 2>&1 | grep -E "probe items|passed|failed"
echo "== rank beyond 2 GiB"; timeout -k 10 600 python -m pytest tests/test_rank_gpu.py -m gpu -x -q -k "over_predict and 7" 2>&1 | tail -5

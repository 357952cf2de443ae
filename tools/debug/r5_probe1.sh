#!/bin/bash
# r5 call 1: where the wide-batch steps spend their time (kernel timelines + in-kernel chain timelines), C5 whole on one GPU
set -u
mkdir -p gpurun_out/r5
o=gpurun_out/r5
bash tools/debug/fit_timeline.sh --items 4587 --cond-inc 300 --batch 1000 > $o/tl_c4.txt 2>&1
echo "c4 timeline done"
bash tools/debug/fit_timeline.sh --batch 512 > $o/tl_b512.txt 2>&1
echo "b512 timeline done"
CT_N=4587 CT_B=1000 CT_COND=300 AAE_CHAIN_TS=1 python tools/debug/chain_ts_shape.py 2> $o/cts_c4_1000.txt
CT_N=4587 CT_B=100 CT_COND=300 AAE_CHAIN_TS=1 python tools/debug/chain_ts_shape.py 2> $o/cts_c4_100.txt
CT_N=100000 CT_B=512 AAE_CHAIN_TS=1 python tools/debug/chain_ts_shape.py 2> $o/cts_c3_512.txt
echo "chain ts done"
timeout -k 10 300 python bench.py --items 2200000 --hidden 200 --batch 512 --median-len 60 --steps 10 --warmup 2 --no-cpu --no-extra > $o/c5_world1.json 2> $o/c5_world1.err
echo "c5 rc=$?"
timeout -k 10 400 python bench.py --steps 100 --warmup 10 --no-cpu > $o/bench0.json 2> $o/bench0.err
echo "bench rc=$?"
tail -c 600 $o/c5_world1.json

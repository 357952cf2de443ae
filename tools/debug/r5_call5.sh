#!/bin/bash
set -u
o=gpurun_out/r5; mkdir -p $o
CT_N=4587 CT_B=1000 CT_COND=300 AAE_CHAIN_TS=1 python tools/debug/chain_ts_shape.py 2> $o/cts16_c4_1000.txt
grep "chain16x3" $o/cts16_c4_1000.txt | tail -4

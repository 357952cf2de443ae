#!/bin/bash
# Short single-GPU runs of bench.py at the shapes of BASELINE.json's configs (numbers quoted in DESIGN.md).
set -u
run() { echo "== $1"; shift; timeout 600 python bench.py --no-cpu --no-extra "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(' docs/s', d['value'], ' ms/step', d['ms_per_step'], ' dominant', d['roofline']['kernel'], d['roofline']['avg_us'], 'us', d['roofline']['bound'], d['roofline']['frac'])"; }
run "C1 shape: N=1k h=50 B=100"            --items 1000 --hidden 50 --batch 100 --steps 200 --warmup 20
run "C2 (bf16): N=47k h=100 B=100"            --dtype bf16 --steps 100 --warmup 10
run "C2 shape (fp32): N=47k h=100 B=100"   --items 47000 --hidden 100 --batch 100 --steps 100 --warmup 10
run "C2 shape (fp32): N=47k h=100 B=500"   --items 47000 --hidden 100 --batch 500 --steps 50 --warmup 5
run "C3: N=100k h=200 B=100"               --items 100000 --hidden 200 --batch 100 --steps 100 --warmup 10
run "C3: N=100k h=200 B=512"               --items 100000 --hidden 200 --batch 512 --steps 50 --warmup 5
run "C4: N=4587 h=200 cond=300 B=1000"     --items 4587 --hidden 200 --batch 1000 --cond-inc 300 --steps 50 --warmup 5
run "C5 shard: N=2.2M h=200 B=64 (1 of 8 ranks' share of B=512)" --items 2200000 --hidden 200 --batch 64 --median-len 60 --steps 20 --warmup 3

#!/bin/bash
# same-box A/B of two TREES: the round-4 worktree (_r4/, its own library and host package) and the working tree
set -u
for i in 1 2 3; do for v in r4 r5; do
  if [ $v = r4 ]; then d=_r4; else d=.; fi
  ( cd $d && AAE_BENCH_EXTRAS=${EXTRAS:-b512,c4} python bench.py --no-cpu --steps 100 --warmup 10 "$@" 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']; e=d.get('extra',{})
print(os.environ['V'], 'C3', d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()}, {n: v.get('ms_per_step') for n,v in e.items()})" )
done; done

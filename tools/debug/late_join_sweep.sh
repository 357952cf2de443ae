# late join (csrc/abi_model.h): headline step time with and without it, over widths of the deferred launch
run() { python bench.py --no-cpu --no-extra --steps 400 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for rep in 1 2; do
  AAE_NO_LATE_JOIN=1 run "early,formula"
  run "late,formula"
  for w in 112 120 128 136 144; do AAE_SPLIT_WGS=$w run "late,$w"; done
done

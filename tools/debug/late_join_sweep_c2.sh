# late join: C2 shape (47 k items, hidden 100), fp32 and bf16, over widths of the deferred launch; plus other shapes at 128
run() { python bench.py --no-cpu --no-extra --steps 400 "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for dt in bf16 f32; do
  A="--dtype $dt --items 47000 --hidden 100"
  AAE_NO_LATE_JOIN=1 run "C2 $dt early,formula" $A
  run "C2 $dt late,formula" $A
  for w in 48 56 64 72 80 96; do AAE_SPLIT_WGS=$w run "C2 $dt late,$w" $A; done
done
for shape in "200000 200" "50000 200" "100000 100" "30000 200"; do
  set -- $shape
  A="--items $1 --hidden $2"
  AAE_NO_LATE_JOIN=1 run "$1x$2 early,formula" $A
  run "$1x$2 late,formula" $A
  for w in 64 96 128; do AAE_SPLIT_WGS=$w run "$1x$2 late,$w" $A; done
done

# bf16 mode, one-term instantiations by default: other shapes, against the three-term kernels
run() { python bench.py --no-cpu --no-extra --steps 300 --dtype bf16 "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"; }
for shape in "47000 100 100" "100000 200 100" "100000 200 512" "20000 200 100" "4587 200 1000"; do
  set -- $shape
  A="--items $1 --hidden $2 --batch $3"
  AAE_NO_BF16_ONE=1 run "$1x$2 b$3 three-term" $A
  run "$1x$2 b$3 one-term" $A
done
for w in 96 112 128 144; do AAE_SPLIT_WGS=$w run "100000x200 b100 one-term, $w" --items 100000 --hidden 200; done

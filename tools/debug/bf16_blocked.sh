# bf16 mode, batches beyond one fused launch on the row-blocked form
run() { python bench.py --no-cpu --no-extra --steps 200 --dtype bf16 "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"; }
for shape in "100000 200 512 0" "4587 200 1000 300" "47000 100 500 0" "100000 200 100 0" "47000 100 100 0"; do
  set -- $shape
  run "$1x$2 b$3 cond$4" --items $1 --hidden $2 --batch $3 --cond-inc $4
done

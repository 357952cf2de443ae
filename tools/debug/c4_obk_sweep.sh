# C4's shape (4 587 items x 1 000 rows, 300-d condition): width of the deferred row-blocked launch
run() { python bench.py --no-cpu --no-extra --steps 200 --items 4587 --hidden 200 --cond-inc 300 --batch 1000 "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"; }
run "f32 default"
for w in 16 32 48 64 96; do AAE_OBK_WGS=$w run "f32 $w"; done
run "bf16 default" --dtype bf16
for w in 16 32 64; do AAE_OBK_WGS=$w run "bf16 $w" --dtype bf16; done

# early prefetch (mark on the step's last launch) vs the mark on the opening gather
run() { python bench.py --no-cpu --no-extra --steps 400 "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
AAE_NO_EARLY_PREFETCH=1 run "C3 mark on the gather"
run "C3 early"
done
AAE_NO_EARLY_PREFETCH=1 run "C2 bf16 gather" --dtype bf16 --items 47000 --hidden 100
run "C2 bf16 early" --dtype bf16 --items 47000 --hidden 100
AAE_NO_EARLY_PREFETCH=1 run "C4 gather" --items 4587 --hidden 200 --cond-inc 300 --batch 1000
run "C4 early" --items 4587 --hidden 200 --cond-inc 300 --batch 1000
AAE_NO_EARLY_PREFETCH=1 run "b512 gather" --batch 512
run "b512 early" --batch 512

# four waves per document in the sparse first layer's gather at wide batches (AAE_GATHER4_ROWS=100000: off)
run() { python bench.py --no-cpu --no-extra --steps 200 "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"; }
for rep in 1 2; do
AAE_GATHER4_ROWS=100000 run "b512 16 waves" --batch 512
run "b512 4 waves" --batch 512
AAE_GATHER4_ROWS=100000 run "C4 16 waves" --items 4587 --hidden 200 --cond-inc 300 --batch 1000
run "C4 4 waves" --items 4587 --hidden 200 --cond-inc 300 --batch 1000
done
AAE_GATHER4_ROWS=100000 run "47k x 500 16 waves" --items 47000 --hidden 100 --batch 500
run "47k x 500 4 waves" --items 47000 --hidden 100 --batch 500
AAE_GATHER4_ROWS=100 run "C3 b100 4 waves (for reference)"
run "C3 b100 16 waves"

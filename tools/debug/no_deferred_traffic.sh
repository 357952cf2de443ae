# TIMING ONLY (wrong results): the step with the deferred launch returning at once (AAE_DEC_SKIP=65536) or holding its CUs
# without touching memory (131072) - what its traffic / its CUs cost the step's own launches
run() { python bench.py --no-cpu --no-extra --steps 400 "${@:2}" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"; }
for rep in 1 2; do
run "normal"
AAE_DEC_SKIP=65536 run "deferred launch empty"
AAE_DEC_SKIP=131072 run "deferred launch holds its CUs for 135 us, no traffic"
done

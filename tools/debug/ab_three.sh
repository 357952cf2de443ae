run() { for i in 1 2; do for v in A B; do
  AAE_HIP_LIB=$PWD/aae-recommender_amd/aaerec/libaaerec_hip_$v.so python bench.py --no-cpu --no-extra "$@" 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']; print(os.environ['V'], d['dtype'], d['value'], d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items() if n.startswith('dec')})"
done; done; }
run --dtype bf16 --steps 100 --warmup 10
run
run --batch 512 --steps 50 --warmup 5

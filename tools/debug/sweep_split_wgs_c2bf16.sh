# as sweep_split_wgs.sh, at C2 with bf16 matrix-core inputs (the deferred launch is dec_fused_bf16_kernel there)
for w in ${WGS:-64 80 96 112 128 160}; do for i in 1 2; do
  export AAE_SPLIT_WGS=$w
  python bench.py --no-cpu --no-extra --dtype bf16 --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print(os.environ['AAE_SPLIT_WGS'], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_us'])"
done; done

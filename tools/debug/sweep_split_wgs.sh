# step time and deferred-launch duration against the deferred launch's workgroup count (batch 100, C3)
for w in ${WGS:-128 136 144 152 160 176}; do for i in 1 2 3; do
  export AAE_SPLIT_WGS=$w
  python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print(os.environ['AAE_SPLIT_WGS'], d['value'], d['ms_per_step'], d['roofline']['avg_us'])"
done; done

# as sweep_split_wgs.sh, at C2's shape (47 k items, hidden 100, fp32, batch 100)
for w in ${WGS:-96 112 128 144 160}; do for i in 1 2 3; do
  export AAE_SPLIT_WGS=$w
  python bench.py --no-cpu --no-extra --items 47000 --hidden 100 --batch 100 --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print(os.environ['AAE_SPLIT_WGS'], d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_us'])"
done; done

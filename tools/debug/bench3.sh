for i in 1 2 3; do python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['value'], d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"; done

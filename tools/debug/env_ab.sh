# same-box A/B of one environment switch of the library: bash tools/debug/env_ab.sh AAE_NO_W1_HYBRID [bench args]
var=$1; shift
for i in 1 2 3 4; do for v in off on; do
  if [ $v = on ]; then export $var=1; else unset $var; fi
  python bench.py --no-cpu --no-extra "$@" 2>/dev/null | tail -1 | V="$var=$v" python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']; print(os.environ['V'], d['value'], d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"
done; done

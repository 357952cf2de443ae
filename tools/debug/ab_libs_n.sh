# same-box A/B of several builds of the library: tools/debug/ab_libs_n.sh "<name> <name> ..." [rounds] [bench args] (libaaerec_hip_<name>.so), alternating runs
NAMES=$1; R=${2:-3}; shift; shift
for i in $(seq 1 $R); do for v in $NAMES; do
  AAE_HIP_LIB=$PWD/aae-recommender_amd/aaerec/libaaerec_hip_$v.so python bench.py --no-cpu --no-extra "$@" 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']; print(os.environ['V'], d['value'], d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()}, flush=True)"
done; done

# extras of bench.py for several builds: tools/debug/extras_ab.sh "<name> ..." (libaaerec_hip_<name>.so)
for v in $1; do
  AAE_HIP_LIB=$PWD/aae-recommender_amd/aaerec/libaaerec_hip_$v.so python bench.py --no-cpu 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print(os.environ['V'], d['ms_per_step'], {k:(v.get('ms_per_step') or v.get('ms_per_call')) for k,v in d['extra'].items()}, flush=True)"
done

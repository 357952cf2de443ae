set -x
L=$PWD/aae-recommender_amd/aaerec
for v in g1pf g1pf2 g13; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so python tools/debug/r6_bits.py 2>&1 | tail -1; done
for v in g1pf2 g13; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so AAE_DEC_TS=x3 python tools/debug/dec_ts.py 2>&1 | grep dec_crit | tail -3; done
for i in 1 2 3; do for v in g1pf g1pf2 g13; do
  AAE_HIP_LIB=$L/libaaerec_hip_$v.so python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']; print(os.environ['V'], d['value'], d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"
done; done 2>&1 | grep -v "^import\|^d=json"

set -x
L=$PWD/aae-recommender_amd/aaerec
for v in pre ""; do AAE_HIP_LIB=$L/libaaerec_hip${v:+_$v}.so python tools/debug/r6_bits.py 2>&1 | tail -1; done
timeout -k 10 900 python -m pytest tests/test_parity_abi_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -5
for i in 1 2 3; do for v in pre ""; do
  AAE_HIP_LIB=$L/libaaerec_hip${v:+_$v}.so python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | V=${v:-new} python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']; print(os.environ['V'], d['value'], d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"
done; done

AAE_FUZZ_SEEDS=112 timeout -k 10 600 python -m pytest tests/test_fuzz_gpu.py -q -k further_activation --tb=line 2>&1 | grep -v "^$" | cut -c1-400 | tail -12
L=$PWD/aae-recommender_amd/aaerec
for v in cur gat2; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so python tools/debug/r6_bits.py 2>&1 | tail -1; done
bash tools/debug/ab_libs2.sh cur gat2 2>&1 | grep -v "^import\|^d=json"
for v in cur gat2; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so python bench.py --no-cpu --no-extra --items 4587 --hidden 200 --batch 1000 --cond-inc 300 --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c4', d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in d['kernels'].items()})"; done

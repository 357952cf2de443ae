AAE_FUZZ_SEEDS=56 timeout -k 10 600 python -m pytest tests/test_fuzz_gpu.py -q -k further_activation --tb=line 2>&1 | grep -v "^\.\|^$" | cut -c1-600 | tail -30
L=$PWD/aae-recommender_amd/aaerec
for v in cur gat; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so python tools/debug/r6_bits.py 2>&1 | tail -1; done
for v in cur gat; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so B=512 python tools/debug/r6_bits.py 2>&1 | tail -1; done
bash tools/debug/ab_libs2.sh cur gat 2>&1 | grep -v "^import\|^d=json"

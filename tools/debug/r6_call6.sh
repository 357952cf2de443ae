set -x
L=$PWD/aae-recommender_amd/aaerec
for v in cur g1pf; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so python tools/debug/r6_bits.py 2>&1 | tail -1; done
for v in cur g1pf; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so B=512 python tools/debug/r6_bits.py 2>&1 | tail -1; done
for v in cur g1pf; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so AAE_DEC_TS=x3 python tools/debug/dec_ts.py 2>&1 | grep dec_crit | tail -3; done
bash tools/debug/ab_libs2.sh cur g1pf 2>&1 | grep -v "^import\|^d=json"

set -x
L=$PWD/aae-recommender_amd/aaerec
for v in base bal j4b; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so python tools/debug/r6_bits.py 2>&1 | tail -1; done
for v in base bal j4b; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so B=512 python tools/debug/r6_bits.py 2>&1 | tail -1; done
for v in bal j4b; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so AAE_DEC_TS=x3 python tools/debug/dec_ts.py 2>&1 | grep dec_crit | tail -5; done
for v in bal j4b; do
AAE_HIP_LIB=$L/libaaerec_hip_$v.so timeout -k 10 600 python -m pytest tests/test_parity_abi_gpu.py -x -q -k "split_output_layer or fused_decoder_equals or late_join" 2>&1 | tail -3
AAE_HIP_LIB=$L/libaaerec_hip_$v.so timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -x -q -k "c3" 2>&1 | tail -3
done
bash tools/debug/ab_libs2.sh base bal 2>&1
bash tools/debug/ab_libs2.sh base j4b 2>&1

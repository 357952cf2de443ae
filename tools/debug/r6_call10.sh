L=$PWD/aae-recommender_amd/aaerec
for v in cur dma; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so python tools/debug/r6_bits.py 2>&1 | tail -1; done
for v in cur dma; do AAE_HIP_LIB=$L/libaaerec_hip_$v.so STEPS=40 python tools/debug/r6_bits.py 2>&1 | tail -1; done
AAE_HIP_LIB=$L/libaaerec_hip_dma.so timeout -k 10 600 python -m pytest tests/test_parity_abi_gpu.py tests/test_fullsize_gpu.py -x -q -k "split_output_layer or late_join or c3 or views_wait" 2>&1 | tail -3
bash tools/debug/ab_libs2.sh cur dma 2>&1 | grep -v "^import\|^d=json"

#!/bin/bash
# every chain program of every test on the wide-batch kernel, then the plain suite
set -u
o=gpurun_out/r5; mkdir -p $o
AAE_X16_ROWS=1 timeout -k 10 900 python -m pytest tests/test_parity_abi_gpu.py tests/test_fuzz_gpu.py tests/test_bf16_gpu.py tests/test_host_gpu.py -m gpu -q > $o/x16_forced.log 2>&1
echo "forced rc=$?"; tail -8 $o/x16_forced.log
timeout -k 10 900 python -m pytest tests -m gpu -q > $o/gpu_tests3.log 2>&1
echo "suite rc=$?"; tail -8 $o/gpu_tests3.log

#!/bin/bash
set -u
o=gpurun_out/r5; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_parity_abi_gpu.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py tests/test_host_gpu.py -m gpu -q -x > $o/slab_direct_tests.log 2>&1
echo "tests rc=$?"; tail -5 $o/slab_direct_tests.log
for rep in 1 2 3; do for v in direct partial; do
  if [ $v = partial ]; then export AAE_NO_SLAB_DIRECT=1; else unset AAE_NO_SLAB_DIRECT; fi
  python bench.py --no-cpu --no-extra --steps 100 --warmup 10 2>/dev/null | tail -1 | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']; print(os.environ['V'], d['value'], d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"
done; done

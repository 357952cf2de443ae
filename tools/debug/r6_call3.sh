set -x
L=$PWD/aae-recommender_amd/aaerec
python tools/debug/r6_bits.py 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests/test_parity_abi_gpu.py -x -q -k "step_act" 2>&1 | tail -15
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "large_batches" 2>&1 | tail -15
timeout -k 10 900 python -m pytest tests/test_parity_abi_gpu.py tests/test_host_gpu.py -x -q 2>&1 | tail -5
python bench.py --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['value'], d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"

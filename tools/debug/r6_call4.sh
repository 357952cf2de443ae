set -x
timeout -k 10 300 python -m pytest tests/test_host_gpu.py -x -q -k "ipc" 2>&1 | tail -15
AAE_BENCH_GLOO_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu 2>gpurun_out/r6_bench_n2.err | tail -1 > gpurun_out/r6_bench_n2.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_n2.json').read())
print({k:d[k] for k in ('value','ms_per_step','n_gpus','dp') if k in d}); print('dp_breakdown',d.get('dp_breakdown')); print('dp_other',d.get('dp_other'))
PY
tail -3 gpurun_out/r6_bench_n2.err
timeout -k 10 500 python bench.py 2>/dev/null | tail -1 > gpurun_out/r6_bench_default.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_default.json').read())
print(d['value'], d['ms_per_step'], d['roofline'], d.get('roofline_critical'), d['cpu_baseline']['value'])
print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('extra',{}).items()})
PY

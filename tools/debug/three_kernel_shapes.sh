#!/bin/bash
# The shapes that take the three-kernel output layer, one line each.  Usage on the GPU box: bash tools/debug/three_kernel_shapes.sh
run() { timeout -k 10 300 python bench.py --no-cpu --no-extra "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('   ms/step', d['ms_per_step'], {n: k[n]['avg_us'] for n in ('dec_bce_fwd', 'dec_da2', 'dec_dv3_adam') if n in k})"; }
echo "== C3 B=512";  run --items 100000 --hidden 200 --batch 512 --steps 30 --warmup 5
echo "== slice 12.5k x 800"; run --items 12500 --hidden 200 --batch 800 --steps 30 --warmup 5
echo "== C4 4587 x 1000"; run --items 4587 --hidden 200 --batch 1000 --cond-inc 300 --steps 30 --warmup 5

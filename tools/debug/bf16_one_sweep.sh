# bf16 mode on the one-term instantiations (AAE_BF16_ONE=1), the deferred launch claiming its CUs' LDS: C2 shape over deferred widths
run() { python bench.py --no-cpu --no-extra --steps 400 --dtype bf16 --items 47000 --hidden 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('$1', d['ms_per_step'], {n: round(v['avg_us'],1) for n,v in k.items()})"; }
run "three-term kernels, formula"
AAE_BF16_ONE=1 AAE_OPT_LDS_KB=1 run "one-term, natural LDS, formula"
AAE_BF16_ONE=1 run "one-term, 150 KB, formula"
for w in 48 56 64 72 96; do AAE_BF16_ONE=1 AAE_SPLIT_WGS=$w run "one-term, 150 KB, $w"; done
AAE_BF16_ONE=1 AAE_OPT_LDS_KB=120 run "one-term, 120 KB, formula"

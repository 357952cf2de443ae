# the first layer's per-item update on wide batches: one wave per item (+ hot list) against one workgroup per item, by vocabulary size
for n in ${NS:-4587 10000 20000 40000}; do for v in default nowave; do
  unset AAE_NO_W1_WAVE
  [ $v = nowave ] && export AAE_NO_W1_WAVE=1
  python bench.py --no-cpu --no-extra --items $n --hidden 200 --batch ${B:-1000} --steps 50 --warmup 5 2>/dev/null | tail -1 | V=$v N=$n python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('N', os.environ['N'], os.environ['V'], d['value'], d['ms_per_step'])"
done; done

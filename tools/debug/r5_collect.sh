#!/bin/bash
# round-5 evidence: tools/collect_profiles.sh r5 (bench lines, kernel stats, PMC passes) + config sweep + step timelines + PMC calibration
set -u
o=gpurun_out; mkdir -p $o
bash tools/collect_profiles.sh r5
echo "[r5] config sweep"; bash tools/config_sweep.sh > $o/r5_config_sweep.log 2>&1
echo "   == C5 WHOLE on one GPU: N=2.2M h=200 B=512 (median length 60)" >> $o/r5_config_sweep.log
timeout 600 python bench.py --no-cpu --no-extra --items 2200000 --hidden 200 --batch 512 --median-len 60 --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(' docs/s', d['value'], ' ms/step', d['ms_per_step'], ' dominant', d['roofline']['kernel'], d['roofline']['avg_us'], 'us', d['roofline']['bound'], d['roofline']['frac'], ' step floor', d['step_roofline']['floor_ms'], 'frac', d['step_roofline']['frac'])" >> $o/r5_config_sweep.log
echo "[r5] timelines"; bash tools/debug/fit_timeline.sh > $o/r5_step_timeline_c3.txt 2>&1
bash tools/debug/fit_timeline.sh --items 4587 --cond-inc 300 --batch 1000 > $o/r5_step_timeline_c4.txt 2>&1
bash tools/debug/fit_timeline.sh --batch 512 > $o/r5_step_timeline_b512.txt 2>&1
echo "[r5] pmc calibration"; bash tools/calib/pmc_calib.sh > /dev/null 2>&1
echo "[r5] done"

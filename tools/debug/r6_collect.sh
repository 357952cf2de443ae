# Round 6 evidence run on the GPU box (two gpurun calls: "a" = bench + kernel stats + counter passes, "b" = timelines, probes, sweep)
set -u
part=${1:-a}
if [ "$part" = a ]; then
  bash tools/collect_profiles.sh r6
else
  echo "[r6] step timeline C3"; bash tools/debug/fit_timeline.sh > gpurun_out/r6_step_timeline_c3.txt 2>&1
  echo "[r6] step timeline b512"; bash tools/debug/fit_timeline.sh --batch 512 > gpurun_out/r6_step_timeline_b512.txt 2>&1
  echo "[r6] critical launch timeline"; AAE_DEC_TS=x3 python tools/debug/dec_ts.py 2>&1 | grep dec_crit > gpurun_out/r6_dec_crit_timeline.txt
  echo "[r6] dp rank compute"; for w in 2 4 8; do for p in 1 2; do python tools/vocab_rank_time.py $w 2>/dev/null | grep "ms/step of compute" | sed "s/^/probe $p: /"; done; done > gpurun_out/r6_dp_rank_compute.log
  echo "[r6] rank rate"; python tools/rank_rate.py > gpurun_out/r6_rank_rate.txt 2>&1
  echo "[r6] config sweep"; bash tools/config_sweep.sh > gpurun_out/r6_config_sweep.log 2>&1
  echo "[r6] done"
fi

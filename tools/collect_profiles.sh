#!/bin/bash
# Round evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag>
# Leaves under gpurun_out/: <tag>_bench.json (default bench.py run), <tag>_kernel_stats.csv (rocprofv3 --kernel-trace
# --stats of the same command), <tag>_bf16_* (the C2 line), pmc_fetch.txt / pmc_write.txt / pmc_mfma.txt (separate
# --pmc passes).  Progress lines go to stdout (a silent run is taken to be hung).
tag=${1:-r4}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root" && mkdir -p gpurun_out
echo "[collect] bench f32"; python3 bench.py > gpurun_out/${tag}_bench.log 2>&1; tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
echo "[collect] bench bf16"; python3 bench.py --dtype bf16 --no-cpu > gpurun_out/${tag}_bf16_bench.log 2>&1; tail -1 gpurun_out/${tag}_bf16_bench.log > gpurun_out/${tag}_bf16_bench.json
for v in f32 bf16; do
  echo "[collect] rocprofv3 kernel stats $v"
  extra=""; [ $v = bf16 ] && extra="--dtype bf16"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/${tag}_prof" -o run -- python3 "$root/bench.py" --no-cpu --no-extra $extra > "$root/gpurun_out/${tag}_prof_$v.log" 2>&1 )
  find gpurun_out/${tag}_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_${v}_kernel_stats.csv \;
  rm -rf gpurun_out/${tag}_prof
done
echo "[collect] pmc fetch"; bash tools/pmc_pass.sh fetch FETCH_SIZE
echo "[collect] pmc write"; bash tools/pmc_pass.sh write WRITE_SIZE
echo "[collect] pmc mfma"; bash tools/pmc_pass.sh mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
echo "[collect] pmc mfma (bf16 ops: the r3 output-layer kernels multiply on the bf16 matrix cores)"; bash tools/pmc_pass.sh mfma_bf16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES
echo "[collect] pmc lds"; bash tools/pmc_pass.sh lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
echo "[collect] pmc scalar / vector instruction counts (r3: the CU's one scalar unit serves its 16 waves)"; bash tools/pmc_pass.sh insts SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES
echo "[collect] done"

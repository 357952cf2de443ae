#!/bin/bash
# Calibration of the L2 fabric read counters on known-size reads (tools/calib/pmc_calib.hip) + the same counters on the bench's
# deferred launch.  Usage on the GPU box: bash tools/calib/pmc_calib.sh   -> gpurun_out/pmc_calib.txt
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$root/gpurun_out/pmc_calib; mkdir -p "$out"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o "$out/pmc_calib" "$root/tools/calib/pmc_calib.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
{
rocprofv3 -L 2>/dev/null | grep -oE "TCC_EA0?_RDREQ[A-Za-z0-9_]*|TCC_BUBBLE[A-Za-z0-9_]*|FETCH_SIZE|TCC_REQ[A-Za-z0-9_]*|TCC_READ[A-Za-z0-9_]*" | sort -u | tr '\n' ' '; echo
for ctr in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_READ_sum TCC_REQ_sum" "TCC_BUBBLE_sum"; do
  echo "== counters: $ctr"
  rm -rf "$out/c"; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/c" -o run -- "$out/pmc_calib" 2>/dev/null | tail -1
  python3 - "$out/c" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print("  ", k, {n: round(sum(v) / len(v), 1) for n, v in c.items()})
PY
  echo "   -- bench.py (C3, batch 100): the deferred and the critical launch"
  rm -rf "$out/b"; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/b" -o run -- python3 "$root/bench.py" --no-cpu --no-extra --steps 20 --warmup 5 > /dev/null 2>&1
  python3 - "$out/b" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "dec_opt" in r["Kernel_Name"] or "dec_crit" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print("  ", k, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", max(len(v) for v in c.values()))
PY
done
} > "$root/gpurun_out/pmc_calib.txt" 2>&1
rm -rf "$out"
cat "$root/gpurun_out/pmc_calib.txt"

#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run (counters in their own run: no trace domains besides
# --kernel-trace).  Usage on the GPU box:  bash tools/pmc_pass.sh <tag> <COUNTER> [<COUNTER> ...]
# Writes gpurun_out/pmc_<tag>.txt: per kernel name, launches and the mean of every counter.
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out" -o run -- python3 "$root/bench.py" --no-cpu --no-extra --steps 20 --warmup 5 > "$out/bench.log" 2>&1 || true
python3 - "$out" <<'PY' > "$root/gpurun_out/pmc_$tag.txt"
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print(k, {c: round(v / n[(k, c)], 1) for c, v in acc[k].items()}, "launches", max(n[(k, c)] for c in acc[k]))
PY
rm -rf "$out"

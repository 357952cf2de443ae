#!/bin/bash
set -u
line() { python bench.py --no-cpu --no-extra --steps 300 --warmup 30 $* 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2 3; do echo "C3 $(line) | C2 bf16 $(line --dtype bf16) | C1 shape $(line --items 1000 --hidden 50)"; done
AAE_DW_TS=90 python bench.py --no-cpu --no-extra --steps 40 --warmup 10 2>&1 | grep -A3 "grouped_dw launch 9[02]" | cut -c1-200
python -m pytest tests -m gpu -q -x -k "parity or fuzz or e2e or metrics or reference or bf16" 2>&1 | tail -2

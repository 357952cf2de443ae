for v in unset 256 0; do
  if [ $v = unset ]; then unset AAE_DW_KSPLIT_ROWS; else export AAE_DW_KSPLIT_ROWS=$v; fi
  echo "AAE_DW_KSPLIT_ROWS=$v"; AAE_DW_TS=61 python bench.py --no-cpu --no-extra --items 4587 --cond-inc 300 --batch 1000 --steps 30 --warmup 10 2>&1 | grep -A1 "grouped_dw launch 61" | cut -c1-200
done

# Round 6 fuzz hunts on the GPU box -> gpurun_out/r6_fuzz_hunt.txt
out=gpurun_out/r6_fuzz_hunt.txt; : > $out
run() { echo "$1 python -m pytest tests/test_fuzz_gpu.py $2" >> $out; env $1 timeout -k 10 900 python -m pytest tests/test_fuzz_gpu.py -q $2 2>&1 | tail -3 >> $out; echo "[fuzz] $1 done"; }
run "AAE_FUZZ_SEEDS=56" "-k further_activation"
run "AAE_FUZZ_SEEDS=200" "-k random_configuration_matches"
run "AAE_SPLIT_ANY=1 AAE_FUZZ_SEEDS=100" "-k random_configuration_matches"
run "AAE_BLOCKED_ANY=1 AAE_FUZZ_SEEDS=100" "-k random_configuration_matches"

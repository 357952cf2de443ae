. tools/debug/sweep_obk_wgs.sh.lib
run C3_B512 "112 128 144 160 192 224" --items 100000 --hidden 200 --batch 512 --steps 50 --warmup 5
run C2_B500 "64 96 128 160 192" --items 47000 --hidden 100 --batch 500 --steps 50 --warmup 5
run C4_B1000 "48 64 96 128 144" --items 4587 --hidden 200 --batch 1000 --cond-inc 300 --steps 50 --warmup 5

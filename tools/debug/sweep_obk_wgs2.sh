. tools/debug/sweep_obk_wgs.sh.lib
run C3_B512 "176 192 208" --items 100000 --hidden 200 --batch 512 --steps 50 --warmup 5
run C2_B500 "112 128 144" --items 47000 --hidden 100 --batch 500 --steps 50 --warmup 5
run C3_B256 "128 160 192 208" --items 100000 --hidden 200 --batch 256 --steps 50 --warmup 5
run C3_B1024 "160 192 208 224" --items 100000 --hidden 200 --batch 1024 --steps 30 --warmup 5

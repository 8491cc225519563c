#!/bin/bash
O=gpurun_out/r03v
mkdir -p $O
timeout -k 10 400 python tools/ab_config.py --config 2 --rounds 4 "xcdsuper=16" "xcdsuper=64" "xcdsuper=256" 2>&1 | grep -v amdgpu.ids | tee $O/ab_xcdsuper_c2.txt
timeout -k 10 400 python tools/ab_config.py --config 4 --rounds 4 "xcdsuper=16" "xcdsuper=64" "xcdsuper=256" 2>&1 | grep -v amdgpu.ids | tee $O/ab_xcdsuper_c4.txt
timeout -k 10 400 python tools/ab_config.py --config 5 --rounds 4 "xcdsuper=16" "xcdsuper=64" "xcdsuper=256" 2>&1 | grep -v amdgpu.ids | tee $O/ab_xcdsuper_c5.txt
timeout -k 10 400 python tools/ab_config.py --config 3 --rounds 3 "xcdsuper=16" "xcdsuper=64" "xcdsuper=256" 2>&1 | grep -v amdgpu.ids | tee $O/ab_xcdsuper_c3.txt
for xs in 16 64 256; do
echo "## xcdsuper=$xs" | tee -a $O/solve_160_xcdsuper.txt
timeout -k 10 300 python tools/async_noise.py 160 4 "xcdsuper=$xs" 2>&1 | grep "3+ 3" | tee -a $O/solve_160_xcdsuper.txt
timeout -k 10 300 python tools/solve_compare.py 160 4 solver=gcr "only=ilu0 async 3 build + 3 apply" "only=ilu0 async 3 build + 5 " xcdsuper=$xs 2>&1 | grep "gcr(" | tee -a $O/solve_160_xcdsuper.txt
done

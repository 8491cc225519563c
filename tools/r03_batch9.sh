#!/bin/bash
O=gpurun_out/r03p
mkdir -p $O
timeout -k 10 400 python tools/ab_config.py --config 2 --rounds 4 "latestore=0" "latestore=1" "latestore=3" "latestore=2" 2>&1 | grep -v amdgpu.ids | tee $O/ab_latestore_256.txt
timeout -k 10 300 python tools/async_noise.py 160 4 "latestore=3" 2>&1 | grep -v amdgpu.ids | tee $O/noise_latestore3.txt
for rep in 1 2; do
echo "## latestore=3 repetition $rep" | tee -a $O/solve_160_latestore3.txt
timeout -k 10 300 python tools/solve_compare.py 160 4 solver=gcr "only=ilu0 async 3 build + 3 apply" "only=ilu0 async 3 build + 5 " "only=sgs async" interleave=0 latestore=3 2>&1 | grep "gcr(" | tee -a $O/solve_160_latestore3.txt
done

#!/bin/bash
O=gpurun_out/r03k
mkdir -p $O
timeout -k 10 300 python tools/async_noise.py 160 4 2>&1 | grep -v amdgpu.ids | tee $O/async_noise_160.txt
for rep in 1 2 3; do
for il in 0 1; do
echo "## interleave=$il repetition $rep" | tee -a $O/solve_160_orders.txt
timeout -k 10 300 python tools/solve_compare.py 160 4 solver=gcr "only=ilu0 async 3 build + 3 apply" "only=ilu0 async 3 build + 5 " interleave=$il 2>&1 | grep "gcr" | tee -a $O/solve_160_orders.txt
done; done

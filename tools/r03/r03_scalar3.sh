#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
rm -f $O/solve_compare_scalar_small.txt
for N in 64 100; do for L in 0 1; do
  timeout -k 10 300 python tools/solve_compare.py $N 1 solver=gcr restart=30 only=apply scalarlane=$L >> $O/solve_compare_scalar_small.txt 2>&1 || exit 1
done; done
grep -v "amdgpu.ids\|DETERMINISTIC" $O/solve_compare_scalar_small.txt

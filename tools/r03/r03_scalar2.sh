#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "scalar_lane" > $O/pytest_scalar2.log 2>&1; rc=$?
echo "pytest scalar rc=$rc"; tail -n 3 $O/pytest_scalar2.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/scalar_ab.py 256 > $O/scalar_ab2.txt 2>&1; echo "scalar_ab rc=$?"
grep -v amdgpu.ids $O/scalar_ab2.txt
rm -f $O/solve_compare_scalar.txt
for L in 4 0 1 4; do
  timeout -k 10 300 python tools/solve_compare.py 160 1 solver=gcr restart=30 only=apply scalarlane=$L >> $O/solve_compare_scalar.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/solve_compare_scalar.txt

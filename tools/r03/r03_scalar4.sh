#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -k "scalar_lane or spmv or gemv or relax" > $O/pytest_scalar4.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -n 3 $O/pytest_scalar4.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/scalar_ab.py 128 256 > $O/scalar_ab4.txt 2>&1; echo "scalar_ab rc=$?"
grep -v amdgpu.ids $O/scalar_ab4.txt | grep -v "factor1plan"

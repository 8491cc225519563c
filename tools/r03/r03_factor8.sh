#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -k "row_path" > $O/pytest_factor8.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -n 3 $O/pytest_factor8.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python tools/factor_rowpath_ab.py bs=8 100 128 > $O/factor_rowpath_ab.txt 2>&1; echo "ab8 rc=$?"
timeout -k 10 400 python tools/factor_rowpath_ab.py bs=4 128 256 >> $O/factor_rowpath_ab.txt 2>&1; echo "ab4 rc=$?"
grep -v amdgpu.ids $O/factor_rowpath_ab.txt

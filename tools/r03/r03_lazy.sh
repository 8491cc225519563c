#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "compact or memory_stats or ilu_apply" > $O/pytest_lazy.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -n 3 $O/pytest_lazy.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python tools/compact_lazy_ab.py > $O/compact_lazy_ab.txt 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids $O/compact_lazy_ab.txt

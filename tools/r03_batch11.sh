#!/bin/bash
O=gpurun_out/r03r
mkdir -p $O
timeout -k 10 300 python tools/invert_ab.py 5 2>&1 | grep -v amdgpu.ids | tee $O/invert_ab.txt
timeout -k 10 300 python tools/invert_ab.py 4 2>&1 | grep -v amdgpu.ids | tee -a $O/invert_ab.txt
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt
tail -n 5 $O/pytest_gpu.log

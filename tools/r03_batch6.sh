#!/bin/bash
set -o pipefail
O=gpurun_out/r03h
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt
tail -n 8 $O/pytest_gpu.log
timeout -k 10 300 python tools/async_quality.py 100 8 interleave=0 interleave=1 2>&1 | grep -v amdgpu.ids | tee $O/async_quality_100_8.txt
timeout -k 10 400 python tools/ab_config.py --config 5 --rounds 4 "interleave=0" "interleave=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_interleave_c5.txt

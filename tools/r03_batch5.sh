#!/bin/bash
set -o pipefail
O=gpurun_out/r03e
mkdir -p $O
timeout -k 10 300 python tools/async_quality.py 128 4 gswave=0 gswave=1 interleave=1 2>&1 | grep -v amdgpu.ids | tee $O/async_quality_128.txt
timeout -k 10 300 python tools/async_quality.py 256 4 gswave=0 gswave=1 interleave=1 2>&1 | grep -v amdgpu.ids | tee $O/async_quality_256.txt
timeout -k 10 400 python tools/ab_config.py --config 2 --rounds 4 "gswave=0" "gswave=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_gswave.txt
python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"
tail -n 4 $O/pytest_parity.log

#!/bin/bash
set -o pipefail
O=gpurun_out/r03f
mkdir -p $O
timeout -k 10 400 python tools/ab_config.py --config 2 --rounds 4 "interleave=0" "interleave=1" "interleave=2" 2>&1 | grep -v amdgpu.ids | tee $O/ab_interleave.txt
timeout -k 10 300 python tools/async_quality.py 256 4 interleave=0 interleave=1 interleave=2 2>&1 | grep -v amdgpu.ids | tee $O/async_quality_256.txt
timeout -k 10 300 python tools/async_quality.py 128 4 interleave=0 interleave=1 interleave=2 2>&1 | grep -v amdgpu.ids | tee $O/async_quality_128.txt



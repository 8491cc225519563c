#!/bin/bash
O=gpurun_out/r03i
mkdir -p $O
timeout -k 10 300 python tools/relax_quality.py 256 4 interleave=0 interleave=3 2>&1 | grep -v amdgpu.ids | tee $O/relax_quality_256.txt
timeout -k 10 300 python tools/relax_quality.py 128 4 interleave=0 interleave=3 2>&1 | grep -v amdgpu.ids | tee $O/relax_quality_128.txt

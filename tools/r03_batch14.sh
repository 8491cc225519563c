#!/bin/bash
O=gpurun_out/r03u
mkdir -p $O
for spec in "64 4" "128 4" "256 4"; do
timeout -k 10 300 python tools/exact_solve_ab.py $spec gatherprobe=0 gatherprobe=5 gatherprobe=6 2>&1 | grep -v amdgpu.ids | tee -a $O/exact_poll_probe.txt || exit 1
done

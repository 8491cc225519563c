#!/bin/bash
O=gpurun_out/r03i
mkdir -p $O
timeout -k 10 400 python tools/ab_config.py --config 2 --rounds 4 "interleave=0;gatherprobe=0" "interleave=1;gatherprobe=0" "interleave=1;gatherprobe=2" "interleave=1;gatherprobe=3" 2>&1 | grep -v amdgpu.ids | tee $O/ab_store_probe.txt

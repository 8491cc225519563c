#!/bin/bash
set -o pipefail
O=gpurun_out/r03c
mkdir -p $O
timeout -k 10 300 python tools/exact_solve_ab.py 256 4 levelpersist=0+levelwide=1+levelnowait=0 levelpersist=0+levelwide=4+levelnowait=0 levelpersist=0+levelwide=1+levelnowait=1 levelpersist=0+levelwide=4+levelnowait=1 2>&1 | grep -v amdgpu.ids | tee -a $O/ab4.txt || exit 1

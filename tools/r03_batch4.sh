#!/bin/bash
set -o pipefail
O=gpurun_out/r03d
mkdir -p $O
timeout -k 10 300 python tools/async_quality.py 256 4 2>&1 | grep -v amdgpu.ids | tee $O/async_quality_256.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 6000 $O/bench_default.json

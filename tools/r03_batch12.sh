#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/pytest_gpu.log
for seed in 31 32 33 34; do
timeout -k 10 400 python tools/fuzz_parity.py 300 $seed 2>&1 | grep -v amdgpu.ids | tail -n 28 | tee -a $O/fuzz.txt
done
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt

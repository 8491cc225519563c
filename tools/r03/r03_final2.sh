#!/bin/bash
# end-of-round validation: fuzz, full GPU suite, smoke, bench
O=gpurun_out/r03z
mkdir -p $O
timeout -k 10 500 python tools/fuzz_parity.py 150 3 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" | tee -a $O/summary.txt; tail -n 12 $O/fuzz.txt
bash tools/r03_final.sh

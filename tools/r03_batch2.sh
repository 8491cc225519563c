#!/bin/bash
set -o pipefail
O=gpurun_out/r03b
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "zero_init or fixed_upper or never_worse" > $O/pytest_new.log 2>&1; echo "pytest parity rc=$?" | tee -a $O/summary.txt
python -m pytest tests/test_gpu_host_api.py tests/test_gpu_petsc.py -q > $O/pytest_host.log 2>&1; echo "pytest host/petsc rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python tools/solve_compare.py 160 4 solver=both "only=ilu0 async" "only=sgs async" "only=seqilu0" "only=sgs exact" "only=sgs DET" > $O/solve_160.txt 2>&1; echo "solve160 rc=$?" | tee -a $O/summary.txt
timeout -k 10 900 python tools/solve_compare.py 256 4 solver=both "only=ilu0 async" "only=sgs async" "only=seqilu0" "only=sgs exact" "only=sgs DET" > $O/solve_256.txt 2>&1; echo "solve256 rc=$?" | tee -a $O/summary.txt
timeout -k 10 1500 bash tools/soak_petsc_driver.sh 1000 $O/r03_soak_petsc.txt > $O/soak.log 2>&1; echo "soak rc=$?" | tee -a $O/summary.txt
tail -n 5 $O/pytest_new.log $O/pytest_host.log

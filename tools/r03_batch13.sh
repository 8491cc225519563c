#!/bin/bash
O=gpurun_out/r03t
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 3 $O/pytest.log
timeout -k 10 400 python tools/ab_config.py --config 5 --rounds 5 "latestore=0" "latestore=2" 2>&1 | grep -v amdgpu.ids | tee $O/ab_latestore_c5.txt
timeout -k 10 300 python tools/async_noise.py 100 8 "latestore=0" "latestore=2" 2>&1 | grep -v amdgpu.ids | tee $O/noise_c5.txt
for rep in 1 2; do for ls in 0 2; do
echo "## bs=8 100^3 latestore=$ls repetition $rep" | tee -a $O/solve_100_8.txt
timeout -k 10 300 python tools/solve_compare.py 100 8 solver=gcr "only=ilu0 async 3 build + 3 apply" "only=ilu0 async 3 build + 5 " interleave=0 latestore=$ls 2>&1 | grep "gcr(" | tee -a $O/solve_100_8.txt
done; done

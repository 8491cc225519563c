#!/bin/bash
O=gpurun_out/r03w
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_levels.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -n 3 $O/pytest.log
timeout -k 10 400 python tools/ab_config.py --config 4 --rounds 5 "latestore=0" "latestore=2" 2>&1 | grep -v amdgpu.ids | tee $O/ab_latestore_c4.txt
timeout -k 10 400 python tools/ab_config.py --config 4 --n 128 --bs 5 --rounds 5 "latestore=0" "latestore=2" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_latestore_c4.txt
timeout -k 10 300 python tools/async_noise.py 100 5 "latestore=0" "latestore=2" 2>&1 | grep -v amdgpu.ids | tee $O/noise_bs5.txt
for rep in 1 2; do for ls in 0 2; do
echo "## bs=5 100^3 latestore=$ls repetition $rep" | tee -a $O/solve_100_5.txt
timeout -k 10 300 python tools/solve_compare.py 100 5 solver=gcr "only=ilu0 async 3 build + 3 apply" "only=ilu0 async 3 build + 5 " latestore=$ls 2>&1 | grep "gcr(" | tee -a $O/solve_100_5.txt
done; done

#!/bin/bash
# round 3: the scalar kernels (kernels_sweep1.hip, factor1p_kernel) -- parity first, then the A/B, the solver-level
# comparison, the profiles of the two kernels at 256^3, then everything (full GPU suite, smoke, bench)
O=gpurun_out/r03s
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "scalar or sync_sweeps or in_place_sweep_variants" > $O/pytest_scalar.log 2>&1; rc=$?
echo "pytest scalar rc=$rc" | tee -a $O/summary.txt; tail -n 3 $O/pytest_scalar.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/scalar_ab.py 128 256 > $O/scalar_ab.txt 2>&1; echo "scalar_ab rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/scalar_ab.txt
for L in 0 1 2; do
  timeout -k 10 300 python tools/solve_compare.py 160 1 gcr 30 only=ilu0 scalarlane=$L >> $O/solve_compare_scalar.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/solve_compare_scalar.txt
export PROF_SKIP_SQ=1
bash tools/profile_round.sh r03_scalar_factor --n 256 --bs 1 --op factor || echo "profile factor failed"
bash tools/profile_round.sh r03_scalar_sweeps --n 256 --bs 1 --op ilu_apply || echo "profile sweeps failed"
bash tools/profile_round.sh r03g_c1 --config 1 || echo "profile c1 failed"
find gpurun_out -name "*_counter_collection.csv" -size +40M -delete -print
bash tools/r03_final.sh

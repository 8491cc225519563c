#!/bin/bash
# end of round 3: refresh the profiles of the configurations whose kernels changed after r03f (config 1: scalar
# kernels, config 5: bs=8 factorisation row path) and the scalar 256^3 runs
export PROF_SKIP_SQ=1
bash /root/repo/tools/profile_round.sh r03h_c1 --config 1 || echo "config 1 failed"
bash /root/repo/tools/profile_round.sh r03h_c5 --config 5 || echo "config 5 failed"
bash /root/repo/tools/profile_round.sh r03_scalar_factor --n 256 --bs 1 --op factor || echo "scalar factor failed"
bash /root/repo/tools/profile_round.sh r03_scalar_spmv --n 256 --bs 1 --op spmv || echo "scalar spmv failed"
find /root/repo/gpurun_out -name "*_counter_collection.csv" -size +40M -delete -print

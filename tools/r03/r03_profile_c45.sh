#!/bin/bash
export PROF_SKIP_SQ=1
bash /root/repo/tools/profile_round.sh r03f_c4 --config 4 || echo "config 4 failed"
bash /root/repo/tools/profile_round.sh r03f_c5 --config 5 || echo "config 5 failed"
find /root/repo/gpurun_out -name "*_counter_collection.csv" -size +40M -delete -print

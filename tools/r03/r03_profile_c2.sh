#!/bin/bash
unset PROF_SKIP_SQ
bash /root/repo/tools/profile_round.sh r03f_c2 --config 2 || echo "config 2 failed"
find /root/repo/gpurun_out -name "*_counter_collection.csv" -size +40M -delete -print

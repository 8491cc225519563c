#!/bin/bash
# end of round 3, final library: the headline configuration and config 4 once more (kernel stats + FETCH/WRITE passes)
export PROF_SKIP_SQ=1
bash /root/repo/tools/profile_round.sh r03z_c2 --config 2 || echo "config 2 failed"
bash /root/repo/tools/profile_round.sh r03z_c4 --config 4 || echo "config 4 failed"
find /root/repo/gpurun_out -name "*_counter_collection.csv" -size +40M -delete -print

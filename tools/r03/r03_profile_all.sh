#!/bin/bash
# round-3 profile set: kernel stats + FETCH/WRITE (+ SQ / TCC) passes of bench.py for BASELINE configs 2, 3, 4, 5, 1
for c in 2 3 4 5 1; do
  if [ $c = 2 ]; then unset PROF_SKIP_SQ; else export PROF_SKIP_SQ=1; fi
  bash /root/repo/tools/profile_round.sh r03_c$c --config $c || echo "config $c failed"
  echo "config $c profiled"
done
du -sh /root/repo/gpurun_out/prof_r03_* | tail -30
# raw counter CSVs can be large: keep what the summariser needs
find /root/repo/gpurun_out -name "*_counter_collection.csv" -size +40M -delete -print

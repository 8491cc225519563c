#!/bin/bash
# usage: scripts_variants.sh "v1 v2 ..." [extra bench args]   (tuning helper, GPU box only)
mkdir -p gpurun_out
for v in $1; do
  echo "== $v"
  BLASTED_HIP_SWEEP4=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%s value=%.1f sweeps/s total=%.0f GB/s  L=%.3f ms U=%.3f ms  U-frac=%.3f other=%.3f' % ('$v', d['value'], d['achieved_gbps'], r['lower_ms'], r['upper_ms'], r['frac'], r['other_ms_per_step']))
"
done

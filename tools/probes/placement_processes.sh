#!/bin/bash
# GPU box: the upper / lower sweep times of the headline configuration in N FRESH processes (every buffer
# re-allocated by a new process each time): the process-to-process spread VERDICT r01 #6 asks about.
# usage: [PLACEMENT=0|1|2] tools/probes/placement_processes.sh <out.txt> [N=12]   (bench.py's own default is the thorough placement, 2)
OUT=$1; N=${2:-12}
echo "# bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline in $N fresh processes: lower / upper sweep ms, fraction of 8 TB/s of the upper sweep, sweeps/s" > $OUT
for i in $(seq 1 $N); do
  timeout -k 10 200 python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-product-default --live-traffic off ${PLACEMENT:+--placement $PLACEMENT} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('process %2d: lower %.3f ms  upper %.3f ms  frac %.3f  value %.1f' % ($i, r['lower_ms'], r['upper_ms'], r['frac'], d['value']))" >> $OUT || exit 1
done
python - <<PY >> $OUT
import re, statistics
up = [float(m.group(1)) for m in re.finditer(r'upper ([0-9.]+) ms', open('$OUT').read())]
lo = [float(m.group(1)) for m in re.finditer(r'lower ([0-9.]+) ms', open('$OUT').read())]
print('upper: min %.3f median %.3f max %.3f ms, spread (max-min)/median %.1f %%' % (min(up), statistics.median(up), max(up), 100 * (max(up) - min(up)) / statistics.median(up)))
print('lower: min %.3f median %.3f max %.3f ms, spread (max-min)/median %.1f %%' % (min(lo), statistics.median(lo), max(lo), 100 * (max(lo) - min(lo)) / statistics.median(lo)))
PY
tail -3 $OUT

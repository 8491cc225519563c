#!/bin/bash
# GPU box: the headline configuration in N fresh processes with the class-aware placement off / quick / thorough.
# usage: tools/probes/placement_ab.sh <out.txt> [N=4] [variants="1 0 2"]
OUT=$1; N=${2:-4}; VARS=${3:-"1 0 2"}
echo "# bench.py --config 2 --steps 10 --warmup 3 in fresh processes: lower / upper sweep ms, fraction of 8 TB/s of the upper sweep, sweeps/s" > $OUT
for i in $(seq 1 $N); do
  for pl in $VARS; do
    t0=$(date +%s.%N)
    BLASTED_HIP_TRACE_PLACEMENT=1 timeout -k 10 200 python bench.py --placement $pl --no-product-default --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --live-traffic off 2>$OUT.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; pl = d.get('placement', {})
w = pl.get('where', {})
print('placement=$pl process %2d: lower %.3f ms  upper %.3f ms  frac %.3f  value %.1f  | turned down %s unchecked %s | lower copy: %s/%s pieces in ytemp class, %s in r class; upper copy: %s/%s in z class, %s in ytemp class; ytemp in r/z class: %s/%s' % ($i, r['lower_ms'], r['upper_ms'], r['frac'], d['value'], pl.get('turned_down'), pl.get('unchecked'), w.get('lower_in_ytemp_class'), w.get('lower_pieces'), w.get('lower_in_r_class'), w.get('upper_in_z_class'), w.get('upper_pieces'), w.get('upper_in_ytemp_class'), w.get('ytemp_in_r_class'), w.get('ytemp_in_z_class')))" >> $OUT || exit 1
    grep "placed " $OUT.err | head -3 | sed -e "s/.*placed/      placed/" | cut -c1-200 >> $OUT
  done
done
rm -f $OUT.err
cat $OUT

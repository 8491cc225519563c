#!/bin/bash
# GPU box: bench.py over a list of configs without the CPU baseline; one summary line per config.
# usage: tools/quick_bench.sh <outdir> <config> [<config> ...]   (extra bench args through BENCH_ARGS)
set -o pipefail
O=$1; shift
mkdir -p $O
for c in "$@"; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline $BENCH_ARGS > $O/bench_c$c.json 2> $O/bench_c$c.err || { echo "bench config $c failed"; tail -5 $O/bench_c$c.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_c$c.json").read().strip().splitlines()[-1])
r=d["roofline"]; f=d.get("factor",{})
print("config $c %-48s %8.1f sweeps/s  %.3f ms/step  frac %.3f  lower %.3f upper %.3f ms | factor sweep %s ms frac %s | exact apply %s ms" % (d["config"]["workload"], d["value"], d["ms_per_step"], r["frac"], r["lower_ms"], r["upper_ms"], f.get("sweep_ms"), f.get("frac"), d.get("exact_apply",{}).get("ms")))
PY
done

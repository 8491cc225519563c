#!/bin/bash
# round-2 GPU batch 1: full GPU test suite, every BASELINE config through bench.py
set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -3 $O/pytest.log
for c in 2 1 3 4 5; do
  timeout -k 10 400 python bench.py --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err || { echo "bench config $c failed"; tail -5 $O/bench_c$c.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_c$c.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("config $c", d["config"]["workload"], "value %.1f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "frac %.3f" % r["frac"], "lower %.3f upper %.3f ms" % (r["lower_ms"], r["upper_ms"]), "cpu", d.get("cpu_baseline",{}).get("value"), "factor", d.get("factor",{}).get("frac"))
PY
done

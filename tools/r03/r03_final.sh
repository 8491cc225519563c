#!/bin/bash
O=gpurun_out/r03z
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/summary.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03z/bench_default.json').read().strip().splitlines()[-1])
print("value %.1f sweeps/s, ms_per_step %.3f, frac %.3f, kernel_ms %.3f lower %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["lower_ms"]))
for o in d["other_configs"]:
    print(o.get("baseline_config"), round(o.get("value"),1), round(o.get("roofline",{}).get("frac"),3), o.get("roofline",{}).get("kernel_ms"))
print("cpu", d["cpu_baseline"]["value"], "exact", d["exact_apply"]["ms"])
PY

#!/bin/bash
O=gpurun_out/r03q
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt
tail -n 4 $O/pytest_gpu.log
for ls in 4 2; do
echo "## latestore=$ls" | tee -a $O/solve_bcgs_latestore.txt
timeout -k 10 400 python tools/solve_compare.py 160 4 solver=bcgs "only=ilu0 async 3 build + 3 apply" "only=ilu0 async 3 build + 5 " "only=ilu0 async 3 build + 10 apply" "only=sgs async" interleave=0 latestore=$ls 2>&1 | grep "bcgs " | tee -a $O/solve_bcgs_latestore.txt
done
echo "## 256^3 latestore=4" | tee -a $O/solve_bcgs_latestore.txt
timeout -k 10 600 python tools/solve_compare.py 256 4 solver=bcgs "only=ilu0 async 3 build + 3 apply" "only=ilu0 async 3 build + 5 " interleave=0 latestore=4 2>&1 | grep "bcgs " | tee -a $O/solve_bcgs_latestore.txt
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03q/bench_default.json').read().strip().splitlines()[-1])
print("value %.1f sweeps/s, ms_per_step %.3f, frac %.3f, kernel_ms %.3f lower %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["lower_ms"]))
print("alt", {k: d["sweep_order_alternative"][k] for k in ("value","upper_ms","upper_frac","contraction_per_sweep","ms_to_1e-6")})
print("quality", {k: d["quality"].get(k) for k in ("contraction_per_sweep","ms_to_1e-6","distance_after_3+3")})
for o in d["other_configs"]:
    print(o.get("baseline_config"), o.get("value"), o.get("roofline",{}).get("frac"), o.get("roofline",{}).get("kernel_ms"))
PY

#!/bin/bash
# the driver's command (default bench.py) with its wall time, and what the line says about traffic
O=gpurun_out/r03z
mkdir -p $O
T0=$(date +%s)
timeout -k 10 700 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03z/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("value %.1f frac %.3f traffic %.5g (%.4f x algorithmic) src %s" % (d["value"], r["frac"], r["traffic"], r["traffic"]/r["algorithmic_bytes_per_launch"], json.dumps(r["traffic_source"])[:300]))
for o in d["other_configs"]:
    ro=o.get("roofline",{}); print(o.get("baseline_config"), round(o.get("value",0),1), round(ro.get("frac",0),3), ro.get("traffic"), ro.get("traffic_over_algorithmic"), (ro.get("traffic_source") or {}).get("seconds"), o.get("without_event_instrumentation"))
print("cpu", d["cpu_baseline"]["value"]); print("factor", json.dumps(d["factor"])[:420])
PY

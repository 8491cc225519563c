#!/bin/bash
O=gpurun_out/r03s
mkdir -p $O
for A in "" "--config 4" "--config 1" "--config 3" "--op factor"; do
  T0=$(date +%s)
  timeout -k 10 500 python bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/live.json 2> $O/live.err || { echo "bench $A failed"; tail -n 5 $O/live.err; }
  T1=$(date +%s)
  python - <<PY
import json
d=json.loads(open("$O/live.json").read().strip().splitlines()[-1])
r=d["roofline"]; ts=r.get("traffic_source") or {}
print("args '$A' wall $((T1-T0)) s: traffic %s algorithmic %s ratio %s live=%s kernel=%s seconds=%s why=%s" % (r.get("traffic"), r.get("algorithmic_bytes_per_launch"), (r["traffic"]/r["algorithmic_bytes_per_launch"] if r.get("traffic") else None), ts.get("live"), (ts.get("kernel") or "")[:70], ts.get("seconds"), ts.get("why_not_live")))
PY
done

#!/bin/bash
# GPU box: for each tuning string (BLASTED_HIP_TUNING), bench.py timing + one FETCH_SIZE / TCC pass.
# usage: tools/ab_fetch.sh <outdir> "<bench args>" <tuning> [<tuning> ...]   ("default" = no tuning)
O=/root/repo/$1; shift
ARGS=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  tag=$(echo "$t" | tr -c 'a-zA-Z0-9\n' '_')
  if [ "$t" = "default" ]; then unset BLASTED_HIP_TUNING; else export BLASTED_HIP_TUNING="$t"; fi
  timeout -k 10 300 python3 /root/repo/bench.py --no-cpu-baseline --live-traffic off $ARGS > $O/bench_$tag.json 2> $O/bench_$tag.err || { echo "bench failed for $t"; tail -3 $O/bench_$tag.err; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/tcc_$tag -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --live-traffic off $ARGS > /dev/null 2> $O/tcc_$tag.err || echo "tcc pass failed"
  python3 - <<PY
import json, csv, glob, collections
d=json.loads(open("$O/bench_$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("%-28s lower %.3f upper %.3f ms  frac %.3f  value %.1f" % ("$t", r["lower_ms"], r["upper_ms"], r["frac"], d["value"]))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/tcc_$tag/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "bhip::" in row["Kernel_Name"] and ("sweep" in row["Kernel_Name"] or "factor" in row["Kernel_Name"]):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,cs in sorted(agg.items(), key=lambda kv: -len(kv[1].get("TCC_MISS_sum",[]))):
    m={c: sum(v)/len(v) for c,v in cs.items()}
    print("    %-70s n=%3d  miss %.2fM (x128 = %.3f GB)  hit %.2fM  rdreq %.2fM  rdreq32 %.2fM" % (k[:70], len(cs["TCC_MISS_sum"]), m.get("TCC_MISS_sum",0)/1e6, m.get("TCC_MISS_sum",0)*128/1e9, m.get("TCC_HIT_sum",0)/1e6, m.get("TCC_EA0_RDREQ_sum",0)/1e6, m.get("TCC_EA0_RDREQ_32B_sum",0)/1e6))
PY
done

#!/bin/bash
# Is the bs=5 factorisation sweep (factorodd_kernel, block products through LDS tiles) bound by the LDS?  Counter
# passes (never combined with a trace) of bench.py --config 4 --op factor.
O=/root/repo/gpurun_out/r03_lds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_ACTIVE_INST_[A-Z_0-9]*" | sort -u > $O/counter_names.txt
B="python3 /root/repo/bench.py --config 4 --op factor --steps 2 --warmup 1 --no-cpu-baseline --live-traffic off --no-other-configs"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/lds -- $B > /dev/null 2> $O/lds.err || echo "lds pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $O/inst -- $B > /dev/null 2> $O/inst.err || echo "inst pass failed"
python3 - <<'PY'
import csv, glob, collections
for d in ("lds", "inst"):
    fs = glob.glob("/root/repo/gpurun_out/r03_lds/%s/*/*_counter_collection.csv" % d)
    if not fs:
        print(d, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "factorodd" in r["Kernel_Name"] or "invert_blocks" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-24s %.4g (avg of %d launches)" % (c, sum(v) / len(v), len(v)))
PY
find $O -name "*_counter_collection.csv" -size +20M -delete

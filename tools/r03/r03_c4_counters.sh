#!/bin/bash
export BLASTED_HIP_PROBES=1  # gatherprobe exists in the probes build only (make -C blasted_amd/csrc probes)
# Config 4 (unstructured bs=5): where the 1.4x traffic of the sweeps comes from.
# Same-process A/B timings of the block-stream policy and of the no-gather probe, then rocprofv3 counter passes
# (never combined with a trace) of bench.py --config 4 for the default, nt1 and the probe.
O=/root/repo/gpurun_out/r03_c4
mkdir -p $O
cd /root/repo
timeout -k 10 400 python tools/ab_config.py --config 4 --rounds 4 "sweepodd=nt0;gatherprobe=0;xcdsuper=16" "sweepodd=nt1;gatherprobe=0;xcdsuper=16" "sweepodd=nt0;gatherprobe=1;xcdsuper=16" "sweepodd=nt1;gatherprobe=1;xcdsuper=16" "sweepodd=nt0;gatherprobe=0;xcdsuper=64" 2>&1 | grep -v amdgpu.ids > $O/ab_timing.txt
cat $O/ab_timing.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCC_READ[A-Za-z0-9_]*\|TCP_TCC[A-Za-z0-9_]*" | sort -u > $O/counter_names.txt
B="python3 /root/repo/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --live-traffic off"
for v in default nt1 probe; do
  case $v in default) export BLASTED_HIP_TUNING="";; nt1) export BLASTED_HIP_TUNING="sweepodd=nt1";; probe) export BLASTED_HIP_TUNING="gatherprobe=1";; esac
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${v}_fetch -- $B > /dev/null 2> $O/${v}_fetch.err || echo "$v fetch failed"
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${v}_write -- $B > /dev/null 2> $O/${v}_write.err || echo "$v write failed"
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/${v}_rdreq -- $B > /dev/null 2> $O/${v}_rdreq.err || echo "$v rdreq failed"
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/${v}_tcc -- $B > /dev/null 2> $O/${v}_tcc.err || echo "$v tcc failed"
  echo "$v done"
done
unset BLASTED_HIP_TUNING
python3 /root/repo/tools/summarize_c4_counters.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
# raw counter CSVs are large: keep only the summary and the per-pass error logs
find $O -name "*_counter_collection.csv" -size +20M -delete

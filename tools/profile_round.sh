#!/bin/bash
# Runs on the GPU box (gpurun).  usage: tools/profile_round.sh <tag> [bench args, e.g. --config 4]
# Collects rocprofv3 kernel stats and, in separate passes (never combined with a trace), the HBM PMC
# counters and a set of SQ / L2 counters for bench.py; raw output under gpurun_out/prof_<tag>_*,
# summaries are made by tools/summarize_prof.py.
set -e
TAG=$1; shift
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --live-traffic off --no-other-configs --no-product-default "$@" > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}_stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --live-traffic off --no-other-configs --no-product-default "$@" > /dev/null 2> $OUT/prof_${TAG}_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --live-traffic off --no-other-configs --no-product-default "$@" > /dev/null 2> $OUT/prof_${TAG}_write.err
if [ -z "$PROF_SKIP_SQ" ]; then
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_${TAG}_sq -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --live-traffic off --no-other-configs --no-product-default "$@" > /dev/null 2> $OUT/prof_${TAG}_sq.err || echo "SQ pass failed"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/prof_${TAG}_tcc -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --live-traffic off --no-other-configs --no-product-default "$@" > /dev/null 2> $OUT/prof_${TAG}_tcc.err || echo "TCC pass failed"
fi
tail -1 $OUT/prof_${TAG}_bench.json | cut -c1-400

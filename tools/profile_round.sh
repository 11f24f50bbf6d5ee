#!/bin/bash
# Profiles of a round, judged from profiles/: usage  tools/profile_round.sh r04x   (GPU box; writes the summaries to gpurun_out/<tag>_*: copy what is kept to profiles/ (gpurun merges only gpurun_out/ back))
# One rocprofv3 collection per pass; --pmc passes never combined with trace domains other than --kernel-trace.
set -u
TAG=${1:-r04}
O=gpurun_out
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
INF="python3 bench.py --no-cpu-baseline --no-extra-legs --no-train-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ks -- $INF > $O/${TAG}_ks.log 2>&1
python3 tools/rocprof_summary.py $O/${TAG}_ks $O/${TAG}_kernel_stats_default_cmd.md
export EG_BENCH_SHARED_CHIP=1       # the eager profiling passes use the GEMM tile policy of the timed 4-lane configuration
EAGER="python3 bench.py --no-graph --no-concurrent --in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-legs --no-train-legs --no-roofline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- $EAGER > $O/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- $EAGER > $O/${TAG}_write.log 2>&1
python3 tools/traffic_json.py $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_traffic.json 8 > $O/${TAG}_traffic_top.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_mfma -- $EAGER > $O/${TAG}_mfma.log 2>&1
python3 tools/pmc_summary.py $O/${TAG}_mfma > $O/${TAG}_pmc_mfma_busy_raw.txt
for B in 16 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_t$B -- python3 bench.py --train --train-batch $B --no-extra-legs --steps 10 --warmup 3 > $O/${TAG}_t$B.log 2>&1
  python3 tools/rocprof_summary.py $O/${TAG}_t$B $O/${TAG}_train_kernel_stats_b$B.md
done
TR="python3 bench.py --train --train-batch 128 --no-train-graph --no-extra-legs --steps 3 --warmup 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_tfetch -- $TR > $O/${TAG}_tfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_twrite -- $TR > $O/${TAG}_twrite.log 2>&1
python3 tools/train_traffic_json.py $O/${TAG}_tfetch $O/${TAG}_twrite 7 128 $O/${TAG}_train_traffic_b128.json $O/${TAG}_train_traffic_b128.md | head -30
rm -rf $O/${TAG}_ks $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_mfma $O/${TAG}_t16 $O/${TAG}_t128 $O/${TAG}_tfetch $O/${TAG}_twrite
ls -la $O/${TAG}_*

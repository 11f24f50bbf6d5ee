#!/bin/bash
# Training-step profiles of a round (GPU box): usage  tools/profile_train.sh r05k   -> gpurun_out/<tag>_train_*  (copy what is kept to profiles/)
# kernel statistics at 16 and 128 clips (one hipGraph per step) and the memory-side bytes per step (separate --pmc FETCH_SIZE / WRITE_SIZE passes, eager launch).
set -u
TAG=${1:-r05k}; O=gpurun_out
cd "$(dirname "$0")/.." 2>/dev/null || true
export TMPDIR=/tmp
for B in 16 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_t$B -- python3 bench.py --train --train-batch $B --no-extra-legs --steps 10 --warmup 3 > $O/${TAG}_t$B.log 2>&1
  python3 tools/rocprof_summary.py $O/${TAG}_t$B $O/${TAG}_train_kernel_stats_b$B.md
  TR="python3 bench.py --train --train-batch $B --no-train-graph --no-extra-legs --steps 3 --warmup 2"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_tfetch$B -- $TR > $O/${TAG}_tfetch$B.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_twrite$B -- $TR > $O/${TAG}_twrite$B.log 2>&1
  python3 tools/train_traffic_json.py $O/${TAG}_tfetch$B $O/${TAG}_twrite$B 7 $B $O/${TAG}_train_traffic_b$B.json $O/${TAG}_train_traffic_b$B.md | head -6
  rm -rf $O/${TAG}_t$B $O/${TAG}_tfetch$B $O/${TAG}_twrite$B
done
ls -la $O/${TAG}_*

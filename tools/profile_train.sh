set -u
TAG=r04h; O=gpurun_out
cd "$(dirname "$0")/.." 2>/dev/null || true
export TMPDIR=/tmp
for B in 16 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_t$B -- python3 bench.py --train --train-batch $B --no-extra-legs --steps 10 --warmup 3 > $O/${TAG}_t$B.log 2>&1
  python3 tools/rocprof_summary.py $O/${TAG}_t$B $O/${TAG}_train_kernel_stats_b$B.md
done
TR="python3 bench.py --train --train-batch 128 --no-train-graph --no-extra-legs --steps 3 --warmup 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_tfetch -- $TR > $O/${TAG}_tfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_twrite -- $TR > $O/${TAG}_twrite.log 2>&1
python3 tools/train_traffic_json.py $O/${TAG}_tfetch $O/${TAG}_twrite 7 128 $O/${TAG}_train_traffic_b128.json $O/${TAG}_train_traffic_b128.md | head -8
rm -rf $O/${TAG}_t16 $O/${TAG}_t128 $O/${TAG}_tfetch $O/${TAG}_twrite
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err
python3 -c "
import json
d=json.load(open('$O/${TAG}_bench_line.json'))
print(d['value'], d['ms_per_step'], d['train']['b16']['ms_per_step'], d['train']['b128']['ms_per_step'], d['train']['b128']['hbm'], d['extra_legs']['gpu_b1']['latency_ms_median'])
"

# same-box A/B of two builds of the library on the inference legs: tmp_lib_old.so (repo root) against the in-tree build
run() { python bench.py --no-cpu-baseline --no-train-legs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra_legs']; print('$1 headline', d['value'], 'pose', d['pose_rel_l2_vs_cpu_oracle'], 'b1 ms', e['gpu_b1']['latency_ms_median'], e['gpu_b1']['latency_ms_min'], 'b1 pose', e['gpu_b1']['pose_rel_l2_vs_cpu_oracle'])"; }
cp emotiongestures_amd/libemogest_hip.so /tmp/new.so
for i in 1 2; do cp /tmp/new.so emotiongestures_amd/libemogest_hip.so; run new; cp tmp_lib_old.so emotiongestures_amd/libemogest_hip.so; run old; done

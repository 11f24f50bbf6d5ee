cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05u > gpurun_out/r05u_profile_round.log 2>&1
rm -rf gpurun_out/prof_b1n; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b1n -- python3 tools/profile_b1.py > gpurun_out/_b1prof.log 2>&1
python3 tools/rocprof_summary.py gpurun_out/prof_b1n gpurun_out/r05u_kernel_stats_b1.md; rm -rf gpurun_out/prof_b1n
python bench.py 2>gpurun_out/r05u_bench_err.txt | tail -1 > gpurun_out/r05u_bench_line.json
python -c "
import json
d=json.load(open('gpurun_out/r05u_bench_line.json')); e=d.get('extra_legs',{}); t=d.get('train',{})
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'b1', e.get('gpu_b1',{}).get('latency_ms_median'), 'beat_long', e.get('beat_long',{}).get('value'), 'div', e.get('diversity_32',{}).get('value'))
print({k:(v.get('ms_per_step'), v.get('library_launches_per_step')) for k,v in t.items() if isinstance(v,dict)})"
ls gpurun_out/r05u_* | head -30

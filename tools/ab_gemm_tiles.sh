# A/B of the pre-split GEMM tile variants: micro-benchmark, then the 4-lane headline with each variant forced (EG_GEMM_TILE)
set -x
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "presplit" 2>&1 | tail -3
python tools/bench_ops.py gemm bf16x3 > gpurun_out/r04c_gemm_tiles.txt 2>&1; cat gpurun_out/r04c_gemm_tiles.txt
for t in 64 128 64r8 128x64r6 128r4 64 128; do
  EG_GEMM_TILE=$t python bench.py --no-train-legs --no-extra-legs --no-cpu-baseline --steps 40 > gpurun_out/r04c_bench_tile_$t.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r04c_bench_tile_$t.json"))
print("TILE $t", d["value"], d["ms_per_step"], d["pose_rel_l2_vs_cpu_oracle"], d["roofline"]["by_kernel_ms_per_step"]["gemm_presplit_kernel (pre-split X)"])
PY
done
for n in 1 2 6 8; do
  python bench.py --no-train-legs --no-extra-legs --no-cpu-baseline --no-roofline --steps 40 --in-flight $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('IN-FLIGHT $n', d['value'], d['ms_per_step'])"
done

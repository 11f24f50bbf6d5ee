set -x
python -m pytest tests/test_training_types.py -x -q -m gpu 2>&1 | tail -8
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "presplit or layernorm or add_rows" 2>&1 | tail -5
./tools/ldsfill_probe > gpurun_out/r04b_ldsfill.txt 2>&1; cat gpurun_out/r04b_ldsfill.txt
python tools/bench_ops.py gemm bf16x3 > gpurun_out/r04b_gemm_tiles.txt 2>&1; cat gpurun_out/r04b_gemm_tiles.txt
for t in 64 128x64 128x64r3 128; do
  EG_GEMM_TILE=$t python bench.py --no-train-legs --no-extra-legs --no-cpu-baseline --steps 40 > gpurun_out/r04b_bench_tile_$t.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r04b_bench_tile_$t.json"))
print("TILE $t", d["value"], d["ms_per_step"], d["pose_rel_l2_vs_cpu_oracle"], d["roofline"]["by_kernel_ms_per_step"])
PY
done

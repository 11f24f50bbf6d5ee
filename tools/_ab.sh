python -m pytest tests/test_gpu_kernels.py -q -x -k "one_clip_products" > gpurun_out/_t0.txt 2>&1; tail -3 gpurun_out/_t0.txt | cut -c1-250; grep -n "^E " gpurun_out/_t0.txt | head -5
run() { EG_GEMM_SKINNY_ROWS=$1 python bench.py --train --train-batch $2 --steps 30 --warmup 5 --no-extra-legs 2>gpurun_out/_ab_err.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('skinny_rows=$1','b',$2, d.get('ms_per_step'), d.get('library_launches_per_step'), d.get('final_loss'))
except Exception as e: print('fail',$2,e)"; }
for i in 1 2; do for f in 64 1024 4096; do run $f 16; run $f 32; done; done
run 64 128; run 8192 128

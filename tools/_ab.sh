python -m pytest tests/test_gpu_generator.py tests/test_data_parallel.py tests/test_gpu_kernels.py -q -x -m gpu > gpurun_out/_t1.txt 2>&1; tail -2 gpurun_out/_t1.txt
run() { EG_CONV_SPLIT=$1 python bench.py --train --train-batch $2 --steps 30 --warmup 5 --no-extra-legs 2>gpurun_out/_ab_err.txt | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('split=[$1]','b',$2, d.get('ms_per_step'), d.get('library_launches_per_step'), d.get('final_loss'))
except Exception as e: print('fail',$2,e)"; }
for i in 1 2; do for f in 1 ""; do run "$f" 16; run "$f" 32; run "$f" 128; done; done

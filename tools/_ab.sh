timeout 700 python -m pytest tests/test_gpu_loops.py -q -x -m gpu > gpurun_out/_tl.txt 2>&1; grep -n "passed\|failed" gpurun_out/_tl.txt | tail -3; tail -5 gpurun_out/_tl.txt | cut -c1-200

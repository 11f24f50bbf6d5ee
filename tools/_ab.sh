python -m pytest tests -q -x -m gpu > gpurun_out/_tall.txt 2>&1; grep -n "passed\|failed" gpurun_out/_tall.txt | tail -3; grep -n "^E " gpurun_out/_tall.txt | head -8

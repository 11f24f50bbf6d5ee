python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "linear" 2>&1 | tail -3
for t in 0 1 0 1; do
 EG_GLDS_TILE=$t python bench.py --train --train-batch 128 --steps 10 --warmup 3 --no-extra-legs --no-train-dropout 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRAIN b128 glds_tile $t', d['ms_per_step'], d['final_loss'])"
done
python bench.py --train --train-batch 128 --steps 10 --warmup 3 --no-extra-legs --no-train-dropout 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRAIN b128 auto', d['ms_per_step'], d['final_loss'])"
python bench.py --train --train-batch 16 --steps 10 --warmup 3 --no-extra-legs --no-train-dropout 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRAIN b16 auto', d['ms_per_step'], d['final_loss'])"

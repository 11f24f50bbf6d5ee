# A/B: CVAE / text / prior branches of the training step on side streams (EG_TRAIN_SIDE_CVAE=1, default) vs one stream (0)
for b in 16 128; do for s in 0 1 0 1; do
 EG_TRAIN_SIDE_CVAE=$s python bench.py --train --train-batch $b --steps 20 --warmup 3 --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRAIN b$b side_streams $s', d['ms_per_step'], repr(d['final_loss']))"
done; done

run() { python bench.py --train --train-batch $1 --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$2 b$1\", d[\"ms_per_step\"])"; }
cp emotiongestures_amd/libemogest_hip.so /tmp/new.so
for i in 1 2 3; do cp /tmp/new.so emotiongestures_amd/libemogest_hip.so; run 16 new; run 128 new; cp tmp_lib_old.so emotiongestures_amd/libemogest_hip.so; run 16 old; run 128 old; done

#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_train; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_train_native.py -x -q > $OUT/tests.log 2>&1; rc=$?; tail -3 $OUT/tests.log; [ $rc -eq 0 ] || exit 1
for r in 1 2 3; do
  python3 bench.py --mode train --steps 300 --warmup 10 > $OUT/train_$r.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.loads(open('$OUT/train_$r.json').read().strip().splitlines()[-1]);print('train overlap', d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
python3 bench.py --mode train --steps 300 --warmup 10 --train-overlap 0 > $OUT/train_noov.json 2>/dev/null || exit 1
python3 -c "import json;d=json.loads(open('$OUT/train_noov.json').read().strip().splitlines()[-1]);print('train no-overlap', d['ms_per_step'], d['host_enqueue_ms_per_step'])"
python3 bench.py --mode seald-train --steps 50 > $OUT/seald_train.json 2>/dev/null || exit 1
python3 -c "import json;d=json.loads(open('$OUT/seald_train.json').read().strip().splitlines()[-1]);print('seald-train', d['ms_per_step'])"

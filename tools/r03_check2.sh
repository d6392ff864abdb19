#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_check2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err || { tail -5 $OUT/bench_driver_cmd.err; exit 1; }
for v in default throughput; do
  if [ $v = throughput ]; then export SDN_FIELD_VARIANT=t; fi
  timeout -k 10 200 python3 bench.py --steps 384 --no-cpu-baseline --no-secondary > $OUT/b_gf4_$v.json 2>/dev/null || exit 1
  timeout -k 10 200 python3 bench.py --steps 384 --group-frames 1 --no-cpu-baseline --no-secondary > $OUT/b_gf1_$v.json 2>/dev/null || exit 1
done
unset SDN_FIELD_VARIANT
timeout -k 10 200 python3 bench.py --mode train --steps 200 --warmup 10 > $OUT/bench_train.json 2>/dev/null || exit 1
python3 - <<'PY'
import json,glob
d=json.loads(open('gpurun_out/r03_check2/bench_driver_cmd.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeats','frames_rendered','timed_region_s')})
print('roofline', {k:d['roofline'][k] for k in ('frac','whole_job_mfma_frac','avg_points_per_launch','instrumented_steps')})
print('grid', d.get('grid_gather_rate'))
for f in sorted(glob.glob('gpurun_out/r03_check2/b_gf*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], 'frac', round(r.get('frac',0),4), 'whole', round(r.get('whole_job_mfma_frac',0),4), 'reps', d['repeats'])
d=json.loads(open('gpurun_out/r03_check2/bench_train.json').read().strip().splitlines()[-1]); print('train ms/step', d['ms_per_step'])
PY

#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_sweep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $OUT/gpu_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err || { tail -5 $OUT/bench_driver_cmd.err; exit 1; }
echo "driver cmd done"
for gf in 1 2 4; do for ctx in 3 4 5 6; do
  timeout -k 10 200 python3 bench.py --steps 384 --group-frames $gf --contexts $ctx --no-cpu-baseline --no-secondary > $OUT/b_gf${gf}_c${ctx}.json 2>/dev/null || exit 1
done; done
python3 - <<'PY'
import json,glob
d=json.loads(open('gpurun_out/r03_sweep/bench_driver_cmd.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','repeats','frames_rendered','timed_region_s')})
print('roofline', {k:d['roofline'][k] for k in ('frac','whole_job_mfma_frac','avg_points_per_launch','instrumented_steps')})
print('secondary', d.get('roofline_secondary')); print('grid', d.get('grid_gather_rate')); print('cpu', d.get('cpu_baseline'))
for f in sorted(glob.glob('gpurun_out/r03_sweep/b_gf*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], 'frac', round(r.get('frac',0),4), 'whole', round(r.get('whole_job_mfma_frac',0),4))
PY

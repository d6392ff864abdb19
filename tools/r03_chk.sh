#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_chk; mkdir -p $OUT
run() { # tag args
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || return 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', '$*', 'ms/frame', round(d['ms_per_step'],4), 'excl', round(r['frac'],4), 'whole-job', round(r['whole_job_mfma_frac'],4), 'lat', round(d['latency_ms_one_loop_at_a_time'],3))"
}
for r in 1 2; do for k in 4 5 6 8; do run k${k}_$r --steps 384 --contexts $k || exit 1; done; done
for k in 6 8; do GPU_MAX_HW_QUEUES=4 run k${k}_q4 --steps 384 --contexts $k || exit 1; done
for k in 6 8; do run k${k}_f2 --steps 384 --contexts $k --group-frames 2 || exit 1; done

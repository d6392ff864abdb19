#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_sweep2; mkdir -p $OUT
run() { # tag args
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || return 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', '$*', 'ms/frame', round(d['ms_per_step'],4), 'excl', round(r['frac'],4), 'whole-job', round(r['whole_job_mfma_frac'],4), 'lat', round(d['latency_ms_one_loop_at_a_time'],3))"
}
for f in 2 3 4 6 8; do run f$f --steps 384 --group-frames $f || exit 1; done
for f in 4 8; do run k3_f$f --steps 384 --contexts 3 --group-frames $f || exit 1; done
run f4_again --steps 384 || exit 1
run f8_again --steps 384 --group-frames 8 || exit 1
run drv_f4 --steps 20 --warmup 5 || exit 1
run drv_f5 --steps 20 --warmup 5 --group-frames 5 || exit 1
run drv_f10 --steps 20 --warmup 5 --group-frames 10 || exit 1

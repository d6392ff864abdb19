#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_chk; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -2 $OUT/tests.log; [ $rc -eq 0 ] || exit 1
run() { # tag args
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || return 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', '$*', 'ms/frame', round(d['ms_per_step'],4), 'excl', round(r['frac'],4), 'whole-job', round(r['whole_job_mfma_frac'],4), 'lat', round(d['latency_ms_one_loop_at_a_time'],3))"
}
run f4a --steps 384 || exit 1
run f4b --steps 384 || exit 1
run drv --steps 20 --warmup 5 || exit 1
run seq --steps 20 --pipeline 0 --group-frames 1 || exit 1

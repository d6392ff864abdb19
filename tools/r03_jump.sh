#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_jump; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -4 $OUT/tests.log; [ $rc -eq 0 ] || exit 1
run() { # tag env args
  local tag=$1 v=$2; shift 2
  for r in 1 2; do
    SDN_CULL_JUMP=$v python3 bench.py "$@" --no-cpu-baseline > $OUT/b_${tag}_$r.json 2>$OUT/b_${tag}_$r.err || return 1
    python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}_$r.json').read().strip().splitlines()[-1]); s=d.get('roofline_secondary') or {}; print('$tag', 'jump=$v', '$*', 'ms/frame', round(d['ms_per_step'],4), 'whole-job', round(d['roofline']['whole_job_mfma_frac'],4), 'marchers', round(s.get('ms_per_frame',0),4), 'latency', round(d.get('latency_ms_one_loop_at_a_time',0),3))"
  done
}
run f4_off 0 --steps 384 || exit 1
run f4_on 1 --steps 384 || exit 1
run f1_off 0 --steps 384 --group-frames 1 || exit 1
run f1_on 1 --steps 384 --group-frames 1 || exit 1
run seq_off 0 --steps 20 --pipeline 0 --group-frames 1 || exit 1
run seq_on 1 --steps 20 --pipeline 0 --group-frames 1 || exit 1
run lego_off 0 --steps 384 --scene lego --no-secondary || exit 1
run lego_on 1 --steps 384 --scene lego --no-secondary || exit 1

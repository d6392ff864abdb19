#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_chk; mkdir -p $OUT
run() { # tag args
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || return 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', '$*', 'ms/frame', round(d['ms_per_step'],4), 'excl', round(r['frac'],4), 'whole-job', round(r['whole_job_mfma_frac'],4))"
}
run default --steps 384 || exit 1
SDN_RECOMPACT_PCT=95 run p95 --steps 384 || exit 1
SDN_RECOMPACT_PCT=50 run p50 --steps 384 || exit 1
run default2 --steps 384 || exit 1
SDN_RECOMPACT_PCT=95 run p95b --steps 384 || exit 1

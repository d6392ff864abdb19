#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_chk; mkdir -p $OUT
run() { # tag args
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || return 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', '$*', 'ms/frame', round(d['ms_per_step'],4), 'excl', round(r['frac'],4), 'whole-job', round(r['whole_job_mfma_frac'],4), 'lat', round(d['latency_ms_one_loop_at_a_time'],3))"
}
for m in 65536 524288 65536 524288 131072 262144; do SDN_FUSED_COMPACT_MAX=$m run f4_$m --steps 384 || exit 1; done

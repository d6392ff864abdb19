#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_chk; mkdir -p $OUT
run() { # tag args
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || return 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', '$*', 'ms/frame', round(d['ms_per_step'],4), 'whole-job', round(r['whole_job_mfma_frac'],4))"
}
for q in 7 8 9 10 11 12 13 14; do GPU_MAX_HW_QUEUES=$q run k8_q$q --steps 384 --contexts 8 || exit 1; done
for q in 7 8 9 10 11 12; do GPU_MAX_HW_QUEUES=$q run k6_q$q --steps 384 --contexts 6 || exit 1; done
for q in 4 5 6 7 8 9; do SDN_CTX_PRIORITIES=0 GPU_MAX_HW_QUEUES=$q run k4own_q$q --steps 384 || exit 1; done

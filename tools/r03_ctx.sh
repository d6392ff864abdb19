#!/bin/bash
set -o pipefail
OUT=gpurun_out/ctx; mkdir -p $OUT
run() { # tag args
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || return 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); print('$tag', '$*', 'ms/frame', round(d['ms_per_step'],4), 'whole-job', round(d['roofline']['whole_job_mfma_frac'],4))"
}
for k in 2 3 4 6 8; do for f in 1 2 4; do run k${k}_f${f} --steps 384 --contexts $k --group-frames $f || exit 1; done; done
for p in 1 2 4 8; do run p${p}_f2 --steps 384 --pipeline $p --group-frames 2 || exit 1; done

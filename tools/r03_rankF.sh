#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_rankF; mkdir -p $OUT
run() { # tag args
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || { tail -5 $OUT/b_${tag}.err; return 1; }
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', '$*', 'ms/frame', round(d['ms_per_step'],4), 'excl', round(r['frac'],4), 'whole-job', round(r['whole_job_mfma_frac'],4), 'lat', round(d['latency_ms_one_loop_at_a_time'],3), 'reps', d['repeats'])"
}
for f in 5 10 20; do run r8_s20_f$f --emulate-rank-of 8 --steps 20 --warmup 5 --group-frames $f || exit 1; done
for f in 8 12 16; do run r8_s384_f$f --emulate-rank-of 8 --steps 384 --group-frames $f || exit 1; done
for f in 4 5 10; do run r4_s20_f$f --emulate-rank-of 4 --steps 20 --warmup 5 --group-frames $f || exit 1; done
for f in 2 4 5 10; do run r2_s20_f$f --emulate-rank-of 2 --steps 20 --warmup 5 --group-frames $f || exit 1; done
run n1_s20 --steps 20 --warmup 5 || exit 1

#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_recompact2; mkdir -p $OUT
run() { # tag pct min args
  local tag=$1 p=$2 m=$3; shift 3
  for r in 1 2; do
    SDN_RECOMPACT_PCT=$p SDN_RECOMPACT_MIN=$m python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}_$r.json 2>$OUT/b_${tag}_$r.err || return 1
    python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}_$r.json').read().strip().splitlines()[-1]); print('$tag', 'pct=$p min=$m', '$*', 'ms/frame', round(d['ms_per_step'],4), 'whole-job', round(d['roofline']['whole_job_mfma_frac'],4), 'lat', round(d['latency_ms_one_loop_at_a_time'],3))"
  done
}
for p in 90 95 100; do run f4_p$p $p 8192 --steps 384 || exit 1; done
for m in 2048 4096 16384; do run f4_m$m 95 $m --steps 384 || exit 1; done
for p in 95 100; do run f1_p$p $p 8192 --steps 384 --group-frames 1 || exit 1; done
for p in 95 100; do run r8_p$p $p 8192 --emulate-rank-of 8 --group-frames 8 --steps 384 || exit 1; done
run drv_p95 95 8192 --steps 20 --warmup 5 || exit 1
run drv_p50 50 8192 --steps 20 --warmup 5 || exit 1

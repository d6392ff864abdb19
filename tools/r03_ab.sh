#!/bin/bash
# same-box A/B of two builds of the library: seald-nerf_amd/lib/libsdn_hip_prev.so (the commit before) against libsdn_hip.so
set -o pipefail
OUT=gpurun_out/r03_ab; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/seald-nerf_amd/lib
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -2 $OUT/tests.log; [ $rc -eq 0 ] || exit 1
run() { # tag lib args
  local tag=$1 lib=$2; shift 2
  SDN_LIB_PATH=$L/$lib python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}.json 2>$OUT/b_${tag}.err || return 1
  python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', '$lib', '$*', 'ms/frame', round(d['ms_per_step'],4), 'whole-job', round(r['whole_job_mfma_frac'],4), 'lat', round(d['latency_ms_one_loop_at_a_time'],3))"
}
for r in 1 2 3; do for v in libsdn_hip_prev.so libsdn_hip.so; do run f4_${v}_$r $v --steps 384 || exit 1; done; done
for v in libsdn_hip_prev.so libsdn_hip.so; do run f1_$v $v --steps 384 --group-frames 1 || exit 1; done
for v in libsdn_hip_prev.so libsdn_hip.so; do run seq_$v $v --steps 20 --pipeline 0 --group-frames 1 || exit 1; done
for v in libsdn_hip_prev.so libsdn_hip.so; do run r8_$v $v --emulate-rank-of 8 --group-frames 10 --steps 20 --warmup 5 || exit 1; done

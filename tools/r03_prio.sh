#!/bin/bash
# contexts at different stream priorities: does it break the convoys of the pipelined frame stream?
set -o pipefail
OUT=gpurun_out/prio; mkdir -p $OUT
python3 - <<'PY'
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
a, b = ctypes.c_int(), ctypes.c_int()
print("priority range rc", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b)), "least", a.value, "greatest", b.value)
PY
run() { # tag, prios, args
  local tag=$1 pr=$2; shift 2
  for r in 1 2; do
    SDN_CTX_PRIORITIES="$pr" python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/b_${tag}_$r.json 2>$OUT/b_${tag}_$r.err || return 1
    python3 -c "import json,sys; d=json.loads(open('$OUT/b_${tag}_$r.json').read().strip().splitlines()[-1]); print('$tag', '[$pr]', '$*', 'ms/frame', round(d['ms_per_step'],4), 'whole-job', round(d['roofline']['whole_job_mfma_frac'],4))"
  done
}
run base "" --steps 384 || exit 1
run hnnl "-1,0,0,1" --steps 384 || exit 1
run hhll "-1,-1,1,1" --steps 384 || exit 1
run hnln "-1,0,1,0" --steps 384 || exit 1
run hlhl "-1,1,-1,1" --steps 384 || exit 1
run f1_base "" --steps 384 --group-frames 1 || exit 1
run f1_hnnl "-1,0,0,1" --steps 384 --group-frames 1 || exit 1
run f1_hnln "-1,0,1,0" --steps 384 --group-frames 1 || exit 1
run f2_base "" --steps 384 --group-frames 2 || exit 1
run f2_hnnl "-1,0,0,1" --steps 384 --group-frames 2 || exit 1

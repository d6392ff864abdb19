#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_stagger
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 $OUT/gpu_tests.log
[ $rc -ne 0 ] && exit 1
NS=$GRAFT_REPO_ROOT/seald-nerf_amd/lib/libsdn_hip_nostagger.so
for round in 1 2; do for pts in 32768 65536 126976 262144 1048576; do
  echo "stagger    $(python3 tools/field_microbench.py --points $pts --iters 100 2>/dev/null)" | tee -a $OUT/micro.txt
  echo "nostagger  $(SDN_LIB_PATH=$NS python3 tools/field_microbench.py --points $pts --iters 100 2>/dev/null)" | tee -a $OUT/micro.txt
done; done
echo "stagger   frame8 $(python3 tools/field_microbench.py --frame-samples 8 --iters 100 2>/dev/null)" | tee -a $OUT/micro.txt
echo "nostagger frame8 $(SDN_LIB_PATH=$NS python3 tools/field_microbench.py --frame-samples 8 --iters 100 2>/dev/null)" | tee -a $OUT/micro.txt
for v in stagger nostagger; do
  if [ $v = nostagger ]; then export SDN_LIB_PATH=$NS; fi
  for rep in 1 2; do timeout -k 10 200 python3 bench.py --steps 384 --no-cpu-baseline --no-secondary > $OUT/b_${v}_$rep.json 2>/dev/null || exit 1; done
  timeout -k 10 200 python3 bench.py --steps 20 --pipeline 0 --group-frames 1 --no-cpu-baseline --no-secondary > $OUT/b_seq_${v}.json 2>/dev/null || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_stagger/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], 'frac', round(r.get('frac',0),4), 'whole', round(r.get('whole_job_mfma_frac',0),4), 'lat', round(d['latency_ms_one_loop_at_a_time'],3))
PY

#!/bin/bash
# A/B of the cooperative (16 lanes per ray) marcher against the lane-per-ray kernels: GPU tests, then sequential / pipelined bench lines
# and rocprofv3 kernel stats for both (SDN_GROUP_MARCH=0 selects the old kernels).
set -o pipefail
OUT=gpurun_out/r03_march_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $OUT/gpu_tests.log; tail -15 $OUT/gpu_tests.log
[ $rc -ne 0 ] && exit 1
for mode in 1 0; do
  export SDN_GROUP_MARCH=$mode
  timeout -k 10 200 python3 bench.py --steps 20 --pipeline 0 --no-cpu-baseline > $OUT/bench_seq_g$mode.json 2>/dev/null || exit 1
  timeout -k 10 200 python3 bench.py --steps 384 --no-cpu-baseline > $OUT/bench_pipe_g$mode.json 2>/dev/null || exit 1
  timeout -k 10 200 python3 bench.py --steps 384 --group-frames 2 --no-cpu-baseline > $OUT/bench_pipe_gf2_g$mode.json 2>/dev/null || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_seq_g$mode -o p -- python3 bench.py --steps 20 --warmup 3 --pipeline 0 --no-cpu-baseline > $OUT/prof_seq_g$mode.json 2> $OUT/prof_seq_g$mode.err || exit 1
  cp $(find $OUT/prof_seq_g$mode -name "*kernel_stats.csv" | head -1) $OUT/seq_g${mode}_kernel_stats.csv; rm -rf $OUT/prof_seq_g$mode
  echo "mode $mode done"
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_march_ab/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], 'lat %.3f'%d['latency_ms_one_loop_at_a_time'], 'frac', round(r.get('frac',0),4), 'whole', round(r.get('whole_job_mfma_frac',0),4))
PY
head -12 $OUT/seq_g1_kernel_stats.csv; head -12 $OUT/seq_g0_kernel_stats.csv

#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_caller_fixtures.py tests/test_gpu_train_native.py -x -q -s > $OUT/tests.log 2>&1; echo "pytest rc $?"; grep -n "distance\|passed\|failed" $OUT/tests.log | cut -c1-700
for gf in 4 1; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_gf$gf -o p -- python3 bench.py --steps 200 --group-frames $gf --no-cpu-baseline --no-secondary --min-timed-s 0 > $OUT/bench_gf$gf.json 2> $OUT/bench_gf$gf.err || exit 1
  python3 tools/trace_overlap.py $(find $OUT/tr_gf$gf -name "*kernel_trace.csv" | head -1) --window-ms 80 | tee $OUT/overlap_gf$gf.json
  rm -rf $OUT/tr_gf$gf
done
python3 bench.py --mode train --steps 300 --warmup 10 > $OUT/train.json 2>/dev/null; python3 -c "import json;d=json.loads(open('$OUT/train.json').read());print('train',d['ms_per_step'],d['host_enqueue_ms_per_step'])"

#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_trace2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for gf in 4 1; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_gf$gf -o p -- python3 bench.py --steps 200 --group-frames $gf --no-cpu-baseline --no-secondary --min-timed-s 0 > $OUT/bench_gf$gf.json 2> $OUT/bench_gf$gf.err || exit 1
  f=$(find $OUT/tr_gf$gf -name "*kernel_trace.csv" | head -1)
  head -2 $f > $OUT/trace_head_gf$gf.txt
  python3 tools/trace_overlap.py $f --window-ms 80 > $OUT/overlap_gf$gf.json || exit 1
  cat $OUT/overlap_gf$gf.json
  rm -rf $OUT/tr_gf$gf
done

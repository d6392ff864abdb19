#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_frame_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for gf in 1 4; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr$gf -o p -- python3 bench.py --steps 20 --warmup 2 --pipeline 0 --group-frames $gf --no-cpu-baseline --no-secondary --min-timed-s 0 > $OUT/bench$gf.json 2> $OUT/bench$gf.err || exit 1
f=$(find $OUT/tr$gf -name "*kernel_trace.csv" | head -1)
python3 tools/frame_timeline.py $f > $OUT/timeline_gf$gf.txt || exit 1
rm -rf $OUT/tr$gf
done
cat $OUT/timeline_gf1.txt

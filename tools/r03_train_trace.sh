#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_train_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o p -- python3 bench.py --mode train --steps 60 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err || exit 1
f=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 tools/train_timeline.py $f > $OUT/timeline.txt || exit 1
rm -rf $OUT/tr
cat $OUT/timeline.txt

#!/bin/bash
set -o pipefail
OUT=gpurun_out/r03_fs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SDN_FIELD_STREAMS=1 GPU_MAX_HW_QUEUES=6 timeout -k 10 400 python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_frame_time.py -x -q > $OUT/tests_fs1.log 2>&1; echo "pytest fs1 rc $?"; tail -2 $OUT/tests_fs1.log
for fsn in 0 1 2; do for gf in 4 1; do for ctx in 4 6; do
  export SDN_FIELD_STREAMS=$fsn GPU_MAX_HW_QUEUES=$((ctx + 1 + fsn))
  timeout -k 10 200 python3 bench.py --steps 384 --group-frames $gf --contexts $ctx --no-cpu-baseline --no-secondary > $OUT/b_fs${fsn}_gf${gf}_c${ctx}.json 2>/dev/null || exit 1
done; done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_fs/b_fs*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print(f.split('/')[-1], 'ms/step %.4f'%d['ms_per_step'], 'frac', round(r.get('frac',0),4), 'whole', round(r.get('whole_job_mfma_frac',0),4), 'overlapped', round((r.get('overlapped') or {}).get('frac',0),4))
PY
